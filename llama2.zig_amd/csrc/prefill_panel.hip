// prefill_panel.hip -- the GEMM of the batched prompt pass for SHORT chunks (17 ... 64 tokens) of matrices that stream from
// HBM (round 5): weight-streaming bound like the decode mat-vec, so the kernel is built around the W stream and nothing
// else comes into a CU twice.
//
// prefill_skinny.hip's forms give a block 16 (or 2 x 16) rows of W and ALL of K, so every block re-reads the whole X
// ([P, K], from L2) beside its rows: 1 byte of X per 1-2 bytes of W into every CU, and the W stream stalls at 4.4-5.2 TB/s
// (a CU takes in ~13-15 bytes per cycle of W + X together).  Here the product is cut the other way:
//   * K is cut into RANGES of 512 (kPnRange; 256 for chunks of 33 ... 64 tokens; the last one of a row may be shorter:
//     K % 128 == 0).  A block works on ONE range at a time and keeps that range of X -- the "panel", [16 TMS tokens][512]
//     (TMS = 1, 2; [64][256] at TMS = 4) -- resident in LDS; what streams
//     through the CU is W alone, rows of 512-byte pieces, 64 rows per work item.
//   * A work item is (range, 64 rows): wave w of the block's four owns rows 16 w .. 16 w + 15 of it, brings their 128-k
//     stages in through its OWN ring of three 8-KB LDS buffers (direct-to-LDS loads, non-temporal) and multiplies them
//     against the panel on MFMA 16x16x4 (A = X: 16 tokens, B = W: 16 rows).  The waves share nothing but the panel: no
//     barrier in the loop, a wave's loads run two stages (16 KB) ahead of its multiplies ACROSS item boundaries.
//     Items are ordered range-major and dealt in equal contiguous spans to one persistent block per CU, so a block
//     changes its panel at most a few times per launch (barrier, reload, barrier).
//   * The block leaves the range's partial products in part[range][token][row]; a second launch (panel_reduce*) adds
//     the ranges IN RANGE ORDER and runs the epilogue of the product -- RoPE + q / KV-cache rows (main.zig:336-358),
//     residual (:395 / :422), SiLU * mul (:411-416).  Partial traffic is P x N x ranges x 8 bytes: 1-6 % of the W bytes at 16-32 tokens.
// Arithmetic: an output is the sum over ranges, in range order, of MFMA chains in (stage, u, c) order -- a function of K
// alone: not of the rows a rank owns (row-sharded == unsharded, bit for bit), not of P or the token's place in its tile.
//
// LDS: panel 32 / 64 KB (TMS 1 / 2; rows unpadded, float4 slots XOR-swizzled by token & 15 on the SOURCE address of the
// direct loads, like the tile GEMM) + 4 rings x 3 x 8 KB (rows of 128 floats, slots swizzled by row & 15): 128 / 160 KB,
// one block per CU.  A ds_read_b128 is served in four groups of 16 lanes -- {0-3, 12-15, 20-27}, ... : rows j = 0-3, 12-15
// at k slot q and 4-11 at q + 1 -- and slot ^ (row & 15) puts those 16 lanes on the 16 different 16-byte chunks of the
// 256-byte bank row (row & 7, the first form, met each chunk twice: every operand read took 8 LDS cycles, not 4).
#include "prefill_common.h"

namespace l2z {
namespace {

constexpr int kPnStage = 128;                 // k per ring stage, one / two token tiles: a row's piece is 512 bytes (three / four tiles: 64)
constexpr int kPnRange = 512;                 // k per range, chunks of <= 32 tokens: four stages (33 ... 64 tokens: 256)
// longest chunk that takes the kernel: 33 ... 64 tokens run four token tiles against ranges of 256 (MFMA-bound there: the
// tile GEMM's split-K family took 12.5 ms for 40 ... 64 tokens at the 7B shape, this 10.8 ... 11.2)
constexpr int kPanelDefaultMax = 96;   // round 6: five and six token tiles (65 ... 96 tokens; the tile GEMM took 15.0 ... 17.4 ms there, 7B)
// ... and the shortest: up to 16 tokens the short-prompt GEMMs of prefill_skinny.hip are ahead -- their products need no
// second launch (7B shape, whole prefill of 8 / 16 / 24 / 32 tokens: 5.31 / 5.65 / 8.76 / 8.78 ms there, 6.02 / 6.16 /
// 6.92 / 6.98 here; profiles/r05b_prefill_panel_ab.txt)
constexpr int kPanelDefaultMin = 17;
#ifndef L2Z_PN_EXP
#define L2Z_PN_EXP 0   // experiment builds (scripts/panel_exp.sh): 1 no W loads, 2 no MFMA, 4 no operand reads, 8 the round-5a swizzle, 32 operand reads after the step's MFMAs
#endif
constexpr int kPnSwz = (L2Z_PN_EXP & 8) ? 7 : 15;

struct PanelArgs {
    const float *x;   // [P, K], ldx floats per row
    int ldx;
    // up to three row-major [rows, K] matrices (K floats per row) sharing x: the launch's features are their rows, in
    // order (q | k | v; W1 and W3 row-interleaved are ONE matrix of 2 hidden rows: the device layout)
    const float *w0, *w1, *w2;
    int rows0, rows1, rows2;
    int P, K;         // K: whole 128-k stages (the product's K rounded up: pad_k)
    int ldw;          // floats between rows of the matrices (the product's own K)
    float *part;      // [ranges][16 TMS][N]
    int n_groups;     // groups of 64 rows
    int n_items;      // ranges * n_groups
};

template <int KR, int SK>
__device__ __forceinline__ int pn_range_stages(int K, int r)
{
    const int left = K - r * KR;
    return (left < KR ? left : KR) / SK;
}

// TMS token tiles of 16; KR: k per range (the panel is [16 TMS][KR]: 64 KB at (2, 512) and (4, 256)); kPnDepth: ring
// buffers per wave (its loads run kPnDepth - 1 stages ahead); NW waves per block, each with 16 rows of the item's
// 16 NW; SK: k per ring stage.  Two forms: (4 waves, stages of 128 k) where the W stream is the bound -- one and two
// token tiles -- and (8 waves, stages of 64 k: the same LDS, two waves per SIMD) where the MFMA pipe is -- three and
// four tiles: a lone wave per SIMD leaves it idle whenever that wave waits for a stage or an operand read.
template <int TMS, int KR, int kPnDepth, int NW, int SK>
__global__ __launch_bounds__(64 * NW) void prefill_panel(const PanelArgs a)
{
    constexpr int SLOTS = SK / 4;             // float4 slots per row of a stage
    constexpr int RPL = 64 / SLOTS;           // rows per wave-wide load (1 KB)
    constexpr int LOADS = 16 / RPL;           // wave-wide loads per stage
    constexpr int STAGE_FLOATS = 16 * SK;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    float *panel = smem;                                                       // [16 TMS][KR], swizzled
    float *ring = smem + 16 * TMS * KR + wave * (kPnDepth * STAGE_FLOATS);
    const int N = a.rows0 + a.rows1 + a.rows2;
    const int i0 = (int)((long long)blockIdx.x * a.n_items / gridDim.x);
    const int i1 = (int)((long long)(blockIdx.x + 1) * a.n_items / gridDim.x);
    const v4f zero = {0.f, 0.f, 0.f, 0.f};

    // ---- producer side: this wave's loads, one stage at a time, running ahead of its multiplies ----
    // Load i of a stage brings RPL rows of the wave's 16 (lane / SLOTS picks the row, lane % SLOTS the PHYSICAL float4
    // slot, which holds logical slot ^ (row & 15)).  The 16 rows lie in one matrix (rows % 16 == 0), so a lane's source is
    // a wave-uniform row-0 pointer + a per-lane offset that never changes: no per-item pointer tables.
    size_t loff[LOADS];
#pragma unroll
    for (int i = 0; i < LOADS; i++) {
        const int jl = RPL * i + lane / SLOTS;
        loff[i] = (size_t)jl * (size_t)a.ldw + (size_t)(4 * ((lane % SLOTS) ^ (jl & kPnSwz)));
    }
    int p_item = i0, p_st = 0, p_ns = 0, p_buf = 0, issued = 0;
    const float *p_base = a.w0;  // row 0 of the wave's 16 at k = the range's start
    auto producer_item = [&]() {
        const int r = p_item / a.n_groups, g = p_item - r * a.n_groups;
        p_ns = pn_range_stages<KR, SK>(a.K, r);
        int row = g * (16 * NW) + wave * 16;
        row = row < N ? row : N - 16;                        // a wave past the end: the last 16 rows again (never stored)
        const bool s1 = row >= a.rows0, s2 = row >= a.rows0 + a.rows1;
        const float *base = s1 ? a.w1 : a.w0;
        base = s2 ? a.w2 : base;
        row -= s2 ? a.rows0 + a.rows1 : (s1 ? a.rows0 : 0);
        p_base = base + (size_t)row * (size_t)a.ldw + (size_t)r * KR;
    };
    auto issue_one = [&]() {
        if (p_item >= i1) return;
        float *dst = ring + p_buf * STAGE_FLOATS;
        const float *src = p_base + p_st * SK;
#pragma unroll
        for (int i = 0; i < LOADS; i++)
            if (!(L2Z_PN_EXP & 1)) lds_dma16_nt(src + loff[i], dst + i * 256);
        p_buf = p_buf + 1 == kPnDepth ? 0 : p_buf + 1;
        issued++;
        if (++p_st == p_ns) {
            p_st = 0;
            if (++p_item < i1) producer_item();
        }
    };
    // ---- the panel: range r of X, all 16 TMS token rows (tokens past P: the last token; never stored) ----
    auto load_panel = [&](int r) {
        const int klen = pn_range_stages<KR, SK>(a.K, r) * SK;
        constexpr int SPR = KR / 4;                             // float4 slots per token row
        static_assert((16 * TMS * SPR) % 64 == 0, "whole wave-wide loads");
        // wave-wide load t fills float4 slots 64 t .. 64 t + 63 of the panel taken as one flat array (a token row is SPR
        // slots: a load may straddle two rows when KR is not a multiple of 256); lane -> (token row, physical slot)
        for (int t = wave; t < 16 * TMS * SPR / 64; t += NW) {
            const int flat = 64 * t + lane;
            const int tok = flat / SPR, slot = flat - tok * SPR;
            const int logical = slot ^ (tok & kPnSwz);          // (stays inside the row: SPR % 16 == 0)
            int kk = 4 * logical;
            kk = kk < klen ? kk : klen - 4;                     // (a short last range: the piece past its end is never read)
            const float *src = a.x + (size_t)(tok < a.P ? tok : a.P - 1) * (size_t)a.ldx + (size_t)r * KR + kk;
            lds_dma16(src, panel + 256 * t);
        }
    };

    v4f acc[TMS];
#pragma unroll
    for (int tm = 0; tm < TMS; tm++) acc[tm] = zero;
    if (i0 < i1) producer_item();
#pragma unroll
    for (int d = 0; d < kPnDepth - 1; d++) issue_one();
    int c_item = i0, c_st = 0, c_buf = 0, consumed = 0, cur_range = -1;
    while (c_item < i1) {
        const int r = c_item / a.n_groups;
        const int ns = pn_range_stages<KR, SK>(a.K, r);
        if (c_st == 0 && r != cur_range) {  // block-uniform: every wave walks the same items
            __syncthreads();                // nobody reads the old panel any more
            load_panel(r);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the ring's loads in flight land with it)
            __syncthreads();
            cur_range = r;
        } else {
            // stage `consumed` of this wave has landed: what may still fly are the stages issued after it.
            // (vmcnt counts the item-end stores of the partial products as well, and a wave's memory operations
            // return IN ORDER on gfx9: right after an item end the immediates below also wait for those 4 TMS stores
            // and for the stage loads issued before them -- correct, and once per item (64 rows x one range) the
            // run-ahead is shorter by it.  The in-order return is the documented behaviour of loads AND stores that
            // share vmcnt on gfx9 / CDNA; the cost of the item-end stall has not been measured on its own.)
            const int younger = issued - consumed - 1;
            if (kPnDepth >= 4 && younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS) : "memory");
            else if (younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the operand reads of the stage before have retired: its buffer is free
        issue_one();                                        // kPnDepth - 1 stages ahead, into that buffer
        const v4f *wst = (const v4f *)(ring + c_buf * STAGE_FLOATS);
        const v4f *xp = (const v4f *)panel;
        const int sw = j & kPnSwz;
        // Operands of step u + 1 are requested BEFORE the MFMAs of step u (two register sets; the barrier keeps hipcc from
        // sinking them behind those MFMAs again, where it waited for them at once).  What hipcc makes of it -- the reads under
        // the last MFMA of the step before, one wait per two steps -- measured 3-5 % faster than the reads after the MFMAs
        // AND than reads pinned elsewhere among the MFMAs or hand-placed with exact waits (L2Z_PN_EXP 32; EXPERIMENTS R5.2).
        constexpr int U = SK / 16;  // k = 16 u + 4 q + c of the stage, A and B alike
        v4f bq[2], xq[2][TMS];
        auto operands = [&](int u, v4f &b, v4f (&xa)[TMS]) {
            const int slot = 4 * u + q;
            if (L2Z_PN_EXP & 4) {
                b = (v4f){(float)lane, 1.f, 2.f, (float)u};
#pragma unroll
                for (int tm = 0; tm < TMS; tm++) xa[tm] = (v4f){(float)tm, (float)lane, 3.f, (float)c_st};
            } else {
                b = wst[j * SLOTS + (slot ^ sw)];
#pragma unroll
                for (int tm = 0; tm < TMS; tm++) xa[tm] = xp[(16 * tm + j) * (KR / 4) + ((c_st * SLOTS + slot) ^ sw)];
            }
        };
        auto mfmas = [&](int u, int c0, int c1) {
#pragma unroll
            for (int c = c0; c < c1; c++)
#pragma unroll
                for (int tm = 0; tm < TMS; tm++) {
                    if (L2Z_PN_EXP & 2) acc[tm][c] = fmaf(xq[u & 1][tm][c], bq[u & 1][c], acc[tm][c]);
                    else acc[tm] = __builtin_amdgcn_mfma_f32_16x16x4f32(xq[u & 1][tm][c], bq[u & 1][c], acc[tm], 0, 0, 0);
                }
        };
        operands(0, bq[0], xq[0]);
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (L2Z_PN_EXP & 32) {  // the first form: reads after the MFMAs
                mfmas(u, 0, 4);
                if (u + 1 < U) operands(u + 1, bq[(u + 1) & 1], xq[(u + 1) & 1]);
            } else {
                if (u + 1 < U) {
                    operands(u + 1, bq[(u + 1) & 1], xq[(u + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                mfmas(u, 0, 4);
            }
        }
        c_buf = c_buf + 1 == kPnDepth ? 0 : c_buf + 1;
        consumed++;
        if (++c_st == ns) {
            // the item's partial products: lane (j, q), register i holds token 16 tm + 4 q + i of row j
            const int g = c_item - r * a.n_groups;
            const int row = g * (16 * NW) + wave * 16 + j;
            if (row < N) {
                float *po = a.part + ((size_t)r * (16 * TMS)) * (size_t)N + row;
#pragma unroll
                for (int tm = 0; tm < TMS; tm++)
#pragma unroll
                    for (int i = 0; i < 4; i++) po[(size_t)(16 * tm + 4 * q + i) * (size_t)N] = acc[tm][i];
            }
#pragma unroll
            for (int tm = 0; tm < TMS; tm++) acc[tm] = zero;
            c_st = 0;
            c_item++;
        }
    }
}

// ---- second launch: ranges added in range order + the product's epilogue ----
enum PanelMode { PN_STORE = 0, PN_RESID = 1, PN_SWIGLU = 2, PN_QKV = 3 };

struct PanelReduceArgs {
    const float *part;   // [ranges][P16][N]
    int n_ranges, P16, N, P;
    float *out;          // PN_STORE / PN_RESID: [P, ldo]; PN_SWIGLU: [P, ldo] of N / 2 gated values; PN_QKV: q [P, ldo]
    int ldo;
    const float *res;    // PN_RESID: out = res + product
    int ldres;
    // PN_QKV: features [0, nq) -> RoPE -> out; [nq, nq + nkv) -> RoPE -> key-cache row pos0 + token; then the value cache
    int nq, nkv, ldkv, head_size, pos0;
    float *outk, *outv;
    size_t kv_head_stride;
    const float2 *rope;
};

__device__ __forceinline__ v4f pn_sum_ranges(const PanelReduceArgs &a, int t, int f)
{
    const size_t stride = (size_t)a.P16 * (size_t)a.N;
    const float *p = a.part + (size_t)t * (size_t)a.N + f;
    v4f v = *(const v4f *)p;                       // range 0, then 1, ...: one fixed order
    int r = 1;
    for (; r + 8 <= a.n_ranges; r += 8) {          // eight loads in flight, added in range order
        v4f u[8];
#pragma unroll
        for (int i = 0; i < 8; i++) u[i] = *(const v4f *)(p + (size_t)(r + i) * stride);
#pragma unroll
        for (int i = 0; i < 8; i++) { v.x += u[i].x; v.y += u[i].y; v.z += u[i].z; v.w += u[i].w; }
    }
    for (; r < a.n_ranges; r++) {
        const v4f u = *(const v4f *)(p + (size_t)r * stride);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    return v;
}

__device__ __forceinline__ size_t pn_kv_index(const PanelReduceArgs &a, int pos, int f)
{
    return a.kv_head_stride ? (size_t)(f / a.head_size) * a.kv_head_stride + (size_t)pos * (size_t)a.head_size + (size_t)(f % a.head_size)
                            : (size_t)pos * (size_t)a.ldkv + (size_t)f;
}

template <int MODE>
__global__ __launch_bounds__(256) void panel_reduce(const PanelReduceArgs a)
{
    const int t = blockIdx.y;
    const int f = 4 * (blockIdx.x * 256 + threadIdx.x);
    if (f >= a.N) return;
    const v4f v = pn_sum_ranges(a, t, f);
    if (MODE == PN_STORE) {
        *(v4f *)(a.out + (size_t)t * a.ldo + f) = v;
    } else if (MODE == PN_RESID) {
        const v4f r = *(const v4f *)(a.res + (size_t)t * a.ldres + f);
        v4f o;
        o.x = r.x + v.x; o.y = r.y + v.y; o.z = r.z + v.z; o.w = r.w + v.w;   // main.zig:711
        *(v4f *)(a.out + (size_t)t * a.ldo + f) = o;
    } else if (MODE == PN_SWIGLU) {
        // rows 2p, 2p + 1 = W1 row p, W3 row p (main.zig:405-416)
        float2 o;
        o.x = swiglu_merge(v.x, v.y);
        o.y = swiglu_merge(v.z, v.w);
        *(float2 *)(a.out + (size_t)t * a.ldo + (f >> 1)) = o;
    } else {
        const int seg = f >= a.nq + a.nkv ? 2 : f >= a.nq ? 1 : 0;
        const int fl = f - (seg == 2 ? a.nq + a.nkv : seg == 1 ? a.nq : 0);
        v4f o = v;
        if (seg < 2) {  // RoPE on the pairs (fl, fl + 1), (fl + 2, fl + 3)  (main.zig:346-349)
            const int hs = a.head_size, pos = a.pos0 + t;
            const float2 c0 = a.rope[(size_t)pos * (size_t)(hs >> 1) + (size_t)((fl % hs) >> 1)];
            const float2 c1 = a.rope[(size_t)pos * (size_t)(hs >> 1) + (size_t)(((fl + 2) % hs) >> 1)];
            o.x = v.x * c0.x - v.y * c0.y;
            o.y = v.x * c0.y + v.y * c0.x;
            o.z = v.z * c1.x - v.w * c1.y;
            o.w = v.z * c1.y + v.w * c1.x;
        }
        if (seg == 0) {
            *(v4f *)(a.out + (size_t)t * a.ldo + fl) = o;
        } else {  // :354-358 (four consecutive features of one head: head_size % 4 == 0)
            float *c = seg == 1 ? a.outk : a.outv;
            *(v4f *)(c + pn_kv_index(a, a.pos0 + t, fl)) = o;
        }
    }
}

}  // namespace

int prefill_panel_max_tokens()
{
    const int m = tunables().pf_panel_max;
    return m < 0 ? kPanelDefaultMax : (m > 96 ? 96 : m);
}

// k per range for a chunk of tms token tiles -- part of the arithmetic (the ranges are added in range order): a function
// of the chunk's length class only
static int panel_range(int tms)
{
    return tms <= 3 ? kPnRange : 256;
}

// Whether a [P, n_whole] x K product of the WHOLE model takes the panel kernel (a function of the model and the chunk
// length only: a rank's share of the rows takes what the unsharded pass takes).  rows_mult16: every matrix of the
// launch has a multiple of 16 rows ON THIS RANK (a wave's 16 rows lie in one matrix).
bool prefill_panel_shape(long long n_whole, int P, int K, long long widest_whole)
{
    if (tunables().pf_panel == 0) return false;
    if (x3_stream_shape(n_whole, P, (K + 63) / 64 * 64)) return false;   // the stream form of the planes kernel takes it (prefill_gemm.hip)
    const int p_min = kPanelDefaultMin;
    K = (K + kPnStage - 1) / kPnStage * kPnStage;   // the kernel walks whole 128-k stages (pad_k)
    if (P < p_min || P < 1 || P > prefill_panel_max_tokens() || K < kPnRange) return false;
    if (n_whole * (long long)K * 4 <= ((long long)16 << 20)) return false;  // cache-resident matrices keep the short-prompt forms
    // the partial products of the WHOLE model's launch must fit the workspace an unsharded runstate allocates
    // (prefill_host.cpp prefill_alloc: kPanelWsRows rows of its widest launch) -- decided on the whole model, so
    // that a shard, whose share always fits then, never takes a kernel the unsharded pass could not
    const int tms = (P + 15) / 16, kr = panel_range(tms);
    const long long n_ranges = (K + kr - 1) / kr;
    return n_ranges * 16 * tms * n_whole <= (long long)kPanelWsRows * widest_whole;
}

hipError_t launch_prefill_panel(const PanelProduct &p, int n_cus, const SplitKWs *ws, hipStream_t st)
{
    const int N = p.rows0 + p.rows1 + p.rows2;
    const int Ke = pad_k(p.K, kPnStage, p.ldx);
    if ((p.rows0 % 16) || (p.rows1 % 16) || (p.rows2 % 16) || N < 16 || (p.ldx % 4) || Ke < 0) return hipErrorNotSupported;
    if (((uintptr_t)p.x & 15) || ((uintptr_t)p.w0 & 15) || ((uintptr_t)p.w1 & 15) || ((uintptr_t)p.w2 & 15)) return hipErrorNotSupported;
    if (ws == nullptr || ws->part == nullptr) return hipErrorNotSupported;
    const int tms = (p.P + 15) / 16;   // token tiles of 16: 2 ... 6 (up to 16 tokens the short-prompt GEMMs are ahead: kPanelDefaultMin)
    // (four ring buffers per wave -- for two token tiles against ranges of 256 -- measured slower: 20 / 32 tokens 6.72 /
    // 6.82 ms with three, 7.55 / 7.66 with four; profiles/r05c_prefill_panel_ab.txt)
    const int kr = panel_range(tms);
    const int depth = tms <= 2 || tms == 4 ? 3 : 2;
    const int n_ranges = (Ke + kr - 1) / kr;
    if ((size_t)n_ranges * (size_t)(16 * tms) * (size_t)N > ws->part_floats) return hipErrorNotSupported;
    PanelArgs a = {};
    a.x = p.x; a.ldx = p.ldx; a.w0 = p.w0; a.w1 = p.w1 ? p.w1 : p.w0; a.w2 = p.w2 ? p.w2 : p.w0;
    a.rows0 = p.rows0; a.rows1 = p.rows1; a.rows2 = p.rows2; a.P = p.P; a.K = Ke; a.ldw = p.K;
    a.part = ws->part;
    // one / two tiles: 4 waves, stages of 128 k; three / four: 8 waves, stages of 64 k (the same k order: the same bits
    // as four waves would give; 40 / 48 / 64 tokens 9.04 / 9.16 / 11.11 -> 8.86 / 9.02 / 11.09 ms, r05j_panel_waves_ab.txt)
    const bool eight = tms >= 3;
    const int nw = eight ? 8 : 4, sk = eight ? 64 : kPnStage;
    a.n_groups = (N + 16 * nw - 1) / (16 * nw);
    a.n_items = n_ranges * a.n_groups;
    const size_t lds = (size_t)(16 * tms * kr + nw * depth * 16 * sk) * sizeof(float);
    // Forms (LDS = panel + rings, one block per CU of 160 KB):
    //   2 tiles      [64 KB panel of 512 k] + 4 waves x 3 x 8 KB        the W stream is the bound (one tile: never ahead of the
    //                                                                    short-prompt GEMMs, 6.0 vs 5.3-5.6 ms: removed in round 6)
    //   3 tiles      [96 KB of 512 k]  + 8 waves x 2 x 4 KB              round 6: half the partial products of round 5's ranges
    //                                                                    of 256 (40 / 48 tokens 8.51 / 8.75 -> 8.06 / 8.27 ms)
    //   4 tiles      [64 KB of 256 k]  + 8 x 3 x 4 KB                    (ranges of 384 with two buffers -- 96 + 64 KB, 31 % fewer
    //                                                                    partials -- measured SLOWER: 56 / 64 tokens 10.40 / 10.56 ->
    //                                                                    10.99 / 11.11 ms; ranges of 512 on four waves 11.10)
    //   5, 6 tiles   [80 | 96 KB of 256 k] + 8 x 2 x 4 KB                round 6: 65 ... 96 tokens (72 / 80 / 88 / 96 tokens 14.7 /
    //                                                                    14.8 / 17.7 / 17.8 -> 12.4 / 12.5 / 14.4 / 14.6 ms;
    //                                                                    profiles/r06_panel_forms.txt)
    const void *fn = nullptr;
    switch (tms) {
    case 2: fn = (const void *)prefill_panel<2, kPnRange, 3, 4, kPnStage>; break;
    case 3: fn = (const void *)prefill_panel<3, 512, 2, 8, 64>; break;
    case 4: fn = (const void *)prefill_panel<4, 256, 3, 8, 64>; break;
    case 5: fn = (const void *)prefill_panel<5, 256, 2, 8, 64>; break;
    case 6: fn = (const void *)prefill_panel<6, 256, 2, 8, 64>; break;
    default: return hipErrorNotSupported;
    }
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return hipErrorNotSupported;
    }
    const int grid = a.n_items < n_cus ? a.n_items : n_cus;
    void *params[] = {&a};
    e = hipLaunchKernel(fn, dim3(grid), dim3(64 * nw), params, lds, st);
    if (e != hipSuccess) return e;
    // the ranges, in order, and the epilogue
    PanelReduceArgs r = {};
    r.part = ws->part; r.n_ranges = n_ranges; r.P16 = 16 * tms; r.N = N; r.P = p.P;
    r.out = p.out; r.ldo = p.ldo; r.res = p.res; r.ldres = p.ldres;
    r.nq = p.rows0; r.nkv = p.rows1; r.ldkv = p.ldkv; r.head_size = p.head_size; r.pos0 = p.pos0;
    r.outk = p.outk; r.outv = p.outv; r.kv_head_stride = p.kv_head_stride; r.rope = p.rope;
    const dim3 g2((unsigned)((N / 4 + 255) / 256), (unsigned)p.P);
    switch (p.mode) {
    case PANEL_STORE: hipLaunchKernelGGL(panel_reduce<PN_STORE>, g2, dim3(256), 0, st, r); break;
    case PANEL_RESID: hipLaunchKernelGGL(panel_reduce<PN_RESID>, g2, dim3(256), 0, st, r); break;
    case PANEL_SWIGLU: hipLaunchKernelGGL(panel_reduce<PN_SWIGLU>, g2, dim3(256), 0, st, r); break;
    case PANEL_QKV:
        if ((p.head_size % 4) != 0) return hipErrorInvalidValue;
        hipLaunchKernelGGL(panel_reduce<PN_QKV>, g2, dim3(256), 0, st, r);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace l2z
