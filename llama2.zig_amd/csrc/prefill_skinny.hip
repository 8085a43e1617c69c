// prefill_skinny.hip -- the GEMM forms for short prompts (P <= 64 tokens) of the batched prompt pass:
// weight-streaming bound like the decode mat-vec, MFMA 16x16x4 on 16 features x 16 TMS tokens per block.
#include "prefill_common.h"

namespace l2z {
namespace {

// Short prompts (P <= 64): the product is bound by streaming W once, like the decode mat-vec, and
// a 64 x 64 tile leaves most CUs without a block (N / 64 blocks).  Here a block owns 16 features
// and 16 TMS tokens and feeds MFMA 16x16x4.  MFMA 16x16x4 f32 operands:
//   A: lane l holds A[i = l & 15][k = l >> 4]   B: lane l holds B[k = l >> 4][j = l & 15]
//   D: lane l, reg r holds D[i = 4 (l >> 4) + r][j = l & 15]
// (Rounds 1-5 carried three forms: no LDS, a register-staged LDS form, and the direct-to-LDS ring below, the first two
// as fallbacks for K that is not whole 256-k stages.  Since round 6 every product runs the ring over K rounded up to
// whole stages -- at least three -- against zero-padded activation rows: pad_k, prefill_common.h.)
constexpr int kSkBK = 256;

// The operands are brought in by direct-to-LDS loads through a ring of SW stages.  (The register-staged form of
// round 1 kept one 16-KB stage of W in flight per block, and N = 4096 gives one block per CU: 4 MB on the wire chip-wide where the memory
// system needs ~13 MB (8 TB/s x latency) -- W streamed at 3.1 TB/s.  Here every wave-wide load still
// reads 1 KB of ONE row, but it lands in LDS without passing through VGPRs, so SW - 1 stages (48 KB of
// W per CU at SW = 4) are in flight.  X rides in the same ring: vmcnt retires loads in issue order, so
// a shallower X ring would drain the W ring with it.  A wave-wide load is exactly one row, so the row
// pitch is free: 264 floats make the operand reads conflict-free as ds_read_b128 -- lane (j, q) takes
// the float4 at k = 64 wave + 16 u + 4 q of row j and feeds component c to MFMA (u, c), A and B alike.
// The W loads carry the non-temporal policy (a stream one CU reads once; X, which every block re-reads
// from L2, does not).  7B shape, q / k / v / wo (67 MB): 21.6 -> 15.0 us = 4.5 TB/s; 16-token prompt
// 7.67 -> 6.24 ms, 8 tokens 7.17 -> 5.99, 32 tokens 10.0 -> 8.9 (nt alone: 6.75 -> 6.24 at 16).
// Measured on top of this and not kept: X loaded straight into a register ring (inline-asm loads the
// compiler does not wait for) so that two 67-KB blocks fit a CU: +3 %; 8 waves per block: +3 %; each
// block starting at another stage of K (in case the 4096 row streams camp on a few channels): +9 %.
// Neither depth, nor waves per CU, nor the LDS read width moves it further.  What does: fewer X bytes per
// W byte (the paired form below: 4.4 -> 5.2 TB/s of W) -- every CU takes in ~13-15 bytes per cycle of
// W + X together, on all 256 CUs (32 rows of one matrix per block on 128 CUs: 15.6 -> 21.8 us; skipping
// the X rows past a 4-token prompt, duplicates that hit the L1: no change).
// Whole 256-k stages, at least three: the launcher rounds K up (skinny_k).
constexpr int kSkLD2 = kSkBK + 8;

// NW = 2: TWO weight matrices share the X stage -- w1 | w3 with silu(a) * b as the epilogue (EPI =
// G_SWIGLU, main.zig:405-416) or wk | wv into the two caches (EPI = G_QKV, :354-358): X is a third of
// the bytes brought into the CU instead of half, one launch instead of two.
template <int EPI, int TMS, int SW, int NW = 1>
__global__ __launch_bounds__(kPfBlock) void prefill_skinny_dma(const GemmArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int WST = 16 * kSkLD2, XST = 16 * TMS * kSkLD2, ST = NW * WST + XST;  // floats per stage: W rows, then X rows
    constexpr int LPS = 4 * NW + 4 * TMS;  // this wave's loads per stage
    static_assert(SW >= 3 && SW <= 4, "ring depth");
    static_assert(NW == 1 || (NW == 2 && (EPI == G_SWIGLU || EPI == G_QKV)), "paired forms");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    // 1-D grid when there are several token tiles (a.nty > 0): the blocks that read the same 16 rows of W
    // get ids 8 apart -- one XCD, the same moment -- so that W comes from HBM once (prefill_gemm.hip,
    // block -> tile)
    int bx = blockIdx.x, by = blockIdx.y;
    if (a.nty > 0) {
        const int group = 8 * a.nty, g = bx / group, local = bx - g * group;
        by = local >> 3;
        bx = g * 8 + (local & 7);
        if (bx >= a.ntx) return;
    } else if (a.ntx > 0) {
        // one token tile, a matrix that streams: block ids go round-robin over the 8 XCDs, so id b takes group
        // (b % 8) * per + b / 8 -- every XCD sweeps its own eighth of the rows linearly instead of all of them
        // advancing through one window (7B, 4 / 16 tokens: 5.97 / 6.17 -> 5.30 / 5.60 ms with W1 | W3 interleaved)
        const int per = (int)gridDim.x >> 3;
        bx = (bx & 7) * per + (bx >> 3);
        if (bx >= a.ntx) return;  // padding of the last window
    }
    const int n0 = bx * 16, m0 = by * 16 * TMS;
    const v4f zero = {0.f, 0.f, 0.f, 0.f};
    // this wave's rows of every stage: W rows 4 wave .. 4 wave + 3 (of each matrix), X rows likewise per token tile
    const float *wsrc[NW][4], *xsrc[TMS][4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int r = 4 * wave + i;
        wsrc[0][i] = a.w + (size_t)min(n0 + r, a.N - 1) * a.ldw + 4 * lane;
        if (NW == 2) wsrc[NW - 1][i] = a.w2 + (size_t)min(n0 + r, a.N - 1) * a.ldw + 4 * lane;
#pragma unroll
        for (int tm = 0; tm < TMS; tm++) xsrc[tm][i] = a.x + (size_t)min(m0 + 16 * tm + r, a.P - 1) * a.ldx + 4 * lane;
    }
    const bool w_nt = a.nty <= 1;  // a W row read by ONE block: stream it past the caches
    auto issue = [&](int st, int buf) {
        float *ws = smem + buf * ST, *xs = ws + NW * WST;
#pragma unroll
        for (int m = 0; m < NW; m++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (w_nt) lds_dma16_nt(wsrc[m][i] + (size_t)st * kSkBK, ws + m * WST + (4 * wave + i) * kSkLD2);
                else lds_dma16(wsrc[m][i] + (size_t)st * kSkBK, ws + m * WST + (4 * wave + i) * kSkLD2);
            }
#pragma unroll
        for (int tm = 0; tm < TMS; tm++)
#pragma unroll
            for (int i = 0; i < 4; i++) lds_dma16(xsrc[tm][i] + (size_t)st * kSkBK, xs + (16 * tm + 4 * wave + i) * kSkLD2);
    };
    v4f acc[NW][TMS];
#pragma unroll
    for (int m = 0; m < NW; m++)
#pragma unroll
        for (int tm = 0; tm < TMS; tm++) acc[m][tm] = zero;
    const int nst = a.K / kSkBK;  // launcher: nst >= SW - 1
#pragma unroll
    for (int p = 0; p < SW - 1; p++) issue(p, p);
    int buf = 0, nbuf = SW - 1;
    for (int st = 0; st < nst; st++) {
        // stage st has landed (this wave's part): what may still fly are the younger stages already issued
        const int younger = min(SW - 2, nst - 1 - st);
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // every wave's part of stage st is in LDS; stage st - 1 has been multiplied
        if (st + SW - 1 < nst) issue(st + SW - 1, nbuf);  // into the buffer stage st - 1 has just left
        const float *wr = smem + buf * ST + j * kSkLD2 + 64 * wave + 4 * q;
        const float *xr = wr + NW * WST;
#pragma unroll
        for (int u = 0; u < 4; u++) {  // this wave's quarter of the stage
            v4f b[NW], xa[TMS];
#pragma unroll
            for (int m = 0; m < NW; m++) b[m] = *(const v4f *)(wr + m * WST + 16 * u);
#pragma unroll
            for (int tm = 0; tm < TMS; tm++) xa[tm] = *(const v4f *)(xr + 16 * tm * kSkLD2 + 16 * u);
#pragma unroll
            for (int m = 0; m < NW; m++)
#pragma unroll
                for (int c = 0; c < 4; c++)
#pragma unroll
                    for (int tm = 0; tm < TMS; tm++)
                        acc[m][tm] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[tm][c], b[m][c], acc[m][tm], 0, 0, 0);
        }
        buf = buf + 1 == SW ? 0 : buf + 1;
        nbuf = nbuf + 1 == SW ? 0 : nbuf + 1;
    }
    __syncthreads();
    float *red = smem;  // [4 waves][NW][TMS][4][64]
#pragma unroll
    for (int m = 0; m < NW; m++)
#pragma unroll
        for (int tm = 0; tm < TMS; tm++)
#pragma unroll
            for (int r = 0; r < 4; r++) red[(((wave * NW + m) * TMS + tm) * 4 + r) * 64 + lane] = acc[m][tm][r];
    __syncthreads();
    for (int idx = tid; idx < TMS * 256; idx += kPfBlock) {
        const int tm = idx >> 8, r = (idx >> 6) & 3, l = idx & 63;
        float v = red[(((0 * NW + 0) * TMS + tm) * 4 + r) * 64 + l], v2 = 0.0f;
#pragma unroll
        for (int w = 1; w < 4; w++) v += red[(((w * NW + 0) * TMS + tm) * 4 + r) * 64 + l];
        if (NW == 2) {
            v2 = red[(((0 * NW + NW - 1) * TMS + tm) * 4 + r) * 64 + l];
#pragma unroll
            for (int w = 1; w < 4; w++) v2 += red[(((w * NW + NW - 1) * TMS + tm) * 4 + r) * 64 + l];
        }
        const int tok = m0 + 16 * tm + 4 * (l >> 4) + r;
        const int f = n0 + (l & 15);
        if (EPI == G_ROPE || EPI == G_ROPE_CACHE || EPI == G_QKV) {
            const float partner = __shfl_xor(v, 1, 64);  // feature f ^ 1, same token (main.zig:346-349)
            const int hs = a.head_size;
            const int pos = a.pos0 + (tok < a.P ? tok : 0);
            const float2 cs = a.rope[(size_t)pos * (size_t)(hs >> 1) + (size_t)(((f < a.N ? f : 0) % hs) >> 1)];
            v = (f & 1) ? partner * cs.y + v * cs.x : v * cs.x - partner * cs.y;
        }
        if (tok < a.P && f < a.N) {
            if (NW == 2 && EPI == G_SWIGLU) a.out[(size_t)tok * a.ldo + f] = swiglu_merge(v, v2);  // :411-416
            else if (NW == 2) {  // wk | wv: key-cache row (RoPE above), value-cache row
                a.outk[kv_index(a, a.ldkv, a.pos0 + tok, f)] = v;
                a.outv[kv_index(a, a.ldkv, a.pos0 + tok, f)] = v2;
            }
            else if (EPI == G_STORE || EPI == G_ROPE) a.out[(size_t)tok * a.ldo + f] = v;
            else if (EPI == G_RESID) a.out[(size_t)tok * a.ldo + f] = a.res[(size_t)tok * a.ldres + f] + v;
            else if (EPI == G_SWIGLU) a.out[(size_t)tok * a.ldo + f] = swiglu_merge(a.out[(size_t)tok * a.ldo + f], v);
            else a.out[kv_index(a, a.ldo, a.pos0 + tok, f)] = v;
        }
    }
}

// feature groups dealt to the XCDs window by window (see the kernel)?  Matrices that stream from HBM, enough groups
bool skinny_spread(const GemmArgs &a)
{
    const bool streams = (size_t)a.N * (size_t)a.n_scale * (size_t)a.K * sizeof(float) > ((size_t)16 << 20);
    return streams && a.N >= 16 * 64;
}

// K as the ring kernel multiplies it: whole 256-k stages, at least three (its ring runs two stages ahead)
int skinny_k(const GemmArgs &a)
{
    const int ke = pad_k(a.K < 3 * kSkBK ? 3 * kSkBK : a.K, kSkBK, a.ldx);
    return (a.ldx % 4) != 0 ? -1 : ke;
}

template <int EPI, int TMS>
hipError_t skinny_launch_t(const GemmArgs &a, hipStream_t st)
{
    static_assert(TMS <= 2, "one or two token tiles (at 64 tokens the stage would take 83 KB)");
    dim3 grid((a.N + 15) / 16, (a.P + 16 * TMS - 1) / (16 * TMS));
    GemmArgs args = a;
    args.K = skinny_k(a);
    if (args.K < 0) return hipErrorInvalidValue;
    constexpr int SW = TMS == 1 ? 4 : 3;
    const size_t lds = (size_t)SW * (16 + 16 * TMS) * kSkLD2 * sizeof(float);
    const void *fn = (const void *)prefill_skinny_dma<EPI, TMS, SW>;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    dim3 g1 = grid;
    args.ntx = 0; args.nty = 0;
    if (grid.y == 1 && skinny_spread(a)) {
        args.ntx = (int)grid.x;
        g1 = dim3((grid.x + 7) / 8 * 8);
    }
    if (grid.y > 1) {
        args.ntx = (int)grid.x; args.nty = (int)grid.y;
        g1 = dim3((grid.x + 7) / 8 * 8 * grid.y);
    }
    void *params[] = {&args};
    return hipLaunchKernel(fn, g1, dim3(kPfBlock), params, lds, st);
}

// one token tile per block?  (see skinny_launch)
bool skinny_one_tile(const GemmArgs &a)
{
    const bool cached = (size_t)a.N * (size_t)a.n_scale * (size_t)a.K * sizeof(float) <= ((size_t)16 << 20);
    return a.P <= 16 || cached || (a.P > 32 && a.P <= 48);
}

template <int EPI>
hipError_t skinny_launch(const GemmArgs &a, hipStream_t st)
{
    // Token tiles per block (TMS x 16 tokens).  A matrix that stays in the on-die caches is cheapest
    // re-read per 16 tokens (more blocks, more waves per CU) -- the WHOLE matrix counts: a row shard takes
    // what the unsharded pass takes.  One that streams from HBM: one or two token tiles per block, and
    // from 33 tokens several blocks per 16 rows of W, paired on one XCD by the 1-D grid (64 tokens: 17.4 ms
    // with four token tiles in registers -> 15.8; 40 tokens 15.4 -> 14.0: that form is gone).  The one-
    // and two-tile forms sum in the same order.
    const bool one = skinny_one_tile(a);
    return one ? skinny_launch_t<EPI, 1>(a, st) : skinny_launch_t<EPI, 2>(a, st);
}

}  // namespace

// a.w | a.w2 in one launch of the direct-to-LDS form (epi G_SWIGLU: out = silu(X w^T) * (X w2^T);
// G_QKV: RoPE(X w^T) into a.outk's rows pos0 + token, X w2^T into a.outv's).  hipErrorNotSupported when
// the shape takes another short-prompt form: the caller launches the two products separately.
hipError_t launch_prefill_skinny_pair(int epi, const GemmArgs &a, hipStream_t st)
{
    if (!skinny_one_tile(a)) return hipErrorNotSupported;
    constexpr int SW = 3;
    const size_t lds = (size_t)SW * (32 + 16) * kSkLD2 * sizeof(float);
    const void *fn = epi == G_SWIGLU ? (const void *)prefill_skinny_dma<G_SWIGLU, 1, SW, 2>
                   : epi == G_QKV    ? (const void *)prefill_skinny_dma<G_QKV, 1, SW, 2> : nullptr;
    if (fn == nullptr) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    GemmArgs args = a;
    args.K = skinny_k(a);
    if (args.K < 0) return hipErrorInvalidValue;
    dim3 grid((a.N + 15) / 16, (a.P + 15) / 16), g1 = grid;
    args.ntx = 0; args.nty = 0;
    if (grid.y == 1 && skinny_spread(a)) {
        args.ntx = (int)grid.x;
        g1 = dim3((grid.x + 7) / 8 * 8);
    }
    if (grid.y > 1) {
        args.ntx = (int)grid.x; args.nty = (int)grid.y;
        g1 = dim3((grid.x + 7) / 8 * 8 * grid.y);
    }
    void *params[] = {&args};
    return hipLaunchKernel(fn, g1, dim3(kPfBlock), params, lds, st);
}

hipError_t launch_prefill_skinny(int epi, const GemmArgs &a, hipStream_t st)
{
    switch (epi) {
        case G_STORE: return skinny_launch<G_STORE>(a, st);
        case G_RESID: return skinny_launch<G_RESID>(a, st);
        case G_ROPE: return skinny_launch<G_ROPE>(a, st);
        case G_ROPE_CACHE: return skinny_launch<G_ROPE_CACHE>(a, st);
        case G_CACHE: return skinny_launch<G_CACHE>(a, st);
        case G_SWIGLU: return skinny_launch<G_SWIGLU>(a, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace l2z
