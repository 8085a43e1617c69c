// prefill_gemm.hip -- the tile GEMMs of the batched prompt pass.
//
// The reference feeds the prompt one token at a time through transformer()
// (src/main.zig:999-1000): every matrix op is a mat-vec and the logits of
// prompt positions are thrown away.  Here P prompt tokens go through a layer
// together, so each weight matrix is streamed once per chunk instead of once
// per token and the matrix work is a true dense GEMM:
//     C[P, N] = X[P, K] . W[N, K]^T        (W row-major as in the checkpoint)
// on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate,
// bit-for-bit a k-ordered fmaf chain, 157 TFLOP/s peak = the f32 vector rate).
// Results equal the token-by-token path up to summation order (tested within
// the same logit tolerance); the KV cache and RunState end in the same state.
//
// This file: the LDS-tiled MFMA GEMMs with fused epilogues for P > 64 tokens and the public GEMM
// launchers.  prefill_skinny.hip: P <= 64.  prefill_attention.hip: rmsnorm, embedding gather, causal
// attention of a chunk.
#include <cstdlib>
#include <type_traits>

#include "prefill_common.h"

#ifndef L2Z_X3_EXP
#define L2Z_X3_EXP 0   // experiment builds (scripts/x3_exp.sh): 1 no loads in the loop, 2 no MFMAs, 4 no X reads, 8 no W reads / splits, 16 no barrier, 32 / 64 (stream form) no X / no W loads, 128 (stream form) W by the default cache policy
#endif

#ifndef L2Z_X3_NBUF2
#define L2Z_X3_NBUF2 5   // ring depth of the stream form at two token tiles (experiment builds: 3, 4)
#endif

namespace l2z {
// Whether a [P, n_whole] x K product of the WHOLE model multiplies on the bf16 matrix cores (x3_applies), whether it takes
// the stream form of that kernel (functions of the model and the chunk length only), and the stream form's K ranges
// (part of the arithmetic).  K: as the kernels walk it (rounded up to 64).
bool x3_applies(long long n_whole, int K)
{
    const int m = tunables().pf_x3;
    // (cache-resident matrices -- the stories models -- keep the f32 cores: their prefill is launch-bound and the planes
    // cost a launch per product: stories110M, 256 tokens 1.47 -> 1.58 ms)
    return m >= 2 || (m == 1 && n_whole * (long long)K * 4 > ((long long)16 << 20));
}
bool x3_stream_shape(long long n_whole, int P, int K)
{
    const Tunables &t = tunables();
    return x3_applies(n_whole, K) && P >= t.pf_x3_stream_min && P <= 128 && K >= 256;
}
// Features per block of the stream form: chunks of <= 64 tokens (one / two token tiles: <= 128 registers per lane) run SIXTEEN
// waves -- 8 feature groups x 2 k-groups, 256 features -- the longer ones eight (128 features).  A block's stage time is set
// by its waves' instruction streams (a dependent chain of six MFMAs per token tile, the direct-to-LDS requests, the W split),
// not by the memory system: at two token tiles 1800 cycles per stage whether 32 or 256 blocks run (profiles/r06s); four
// waves per SIMD instead of two cover each other, and the planes cross the chip once per 256 features instead of once per 128.
int x3_stream_tile(int P) { return P <= 64 ? 256 : 128; }
int x3_stream_sk(long long n_whole, int P, int K)
{
    // One block per CU (the ring takes the LDS), blocks of a launch equally long: the launch lasts
    // ceil(tiles sk / CUs) rounds of ceil(stages / sk) stages, plus a fixed cost per round (launch, pipeline fill, the partial
    // sums' hand-over: ~15 us = 20 stages' worth; every product of the 7B shape with every range count, 64 and 128 tokens:
    // profiles/r06r_stream_sk_scan.txt -- with 10, W1 | W3 took 4 ranges in 2.7 rounds: 164 against 154 us unsplit at 128 tokens).  The CU count is the part the rule was made for (MI355X: 256), as a constant -- the
    // ranges are part of the arithmetic and must not depend on the device a rank happens to run on.
#ifdef L2Z_X3_SK_FORCE
    { int f = L2Z_X3_SK_FORCE; while (f > 1 && K / 32 / f < 8) f >>= 1; (void)n_whole; return f; }   // experiment builds
#endif
    constexpr long long kCus = 256;
    const long long tile = x3_stream_tile(P), tiles = (n_whole + tile - 1) / tile, stages = K / 32;
    int best = 1;
    long long best_cost = 0;
    for (int sk = 1; sk <= 8; sk *= 2) {   // (1, 2, 4, 8: the ranges share the tile's epilogue in equal parts)
        if (stages / sk < 8) break;
        const long long rounds = (tiles * sk + kCus - 1) / kCus;
        const long long cost = rounds * ((stages + sk - 1) / sk + 20);
        if (sk == 1 || cost < best_cost) { best = sk; best_cost = cost; }
    }
    return best;
}
}  // namespace l2z

namespace l2z {
namespace {

// compute units of the current device (tile choice: does the larger tile still give every CU a block?)
static int g_cus_hint()
{
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            cus = n;
        else
            return 256;
    }
    return cus;
}

// Output tile of the direct-to-LDS kernel for a [P, N] product.  All forms split and order k the same
// way (two k-groups per 64-k stage, MFMA pairing (8s+t, 8s+4+t)), so the choice never changes a bit of
// the result (tests: sharded == unsharded, L2Z_PF_TILE forms equal) -- it is purely a question of
// filling 256 CUs:  cost = tiles in sequence on the busiest CU x tile area / efficiency of the form,
// efficiencies from interleaved A/B runs on the 7B and 110M shapes (larger tiles bring
// fewer bytes per flop into the CU).  The 7B shape at 512 tokens keeps 128 x 64; 65..128-token prompts,
// small models and row shards (N / world features) take the 32-token forms.
enum TileForm { TILE_128x64 = 0, TILE_64x64, TILE_32x64, TILE_32x32, TILE_128x128 };
static TileForm choose_tile(int N, int P, bool pair)
{
    const long long cus = g_cus_hint();
    // tokens, features (of each matrix when paired) per block, relative efficiency
    static const struct { int tok, feat; double eff; } form[5] = {{128, 64, 1.0}, {64, 64, 0.87}, {32, 64, 0.80}, {32, 32, 0.65},
                                                                   {128, 128, 1.12}};
    int best = -1;
    double best_cost = 0.0;
    for (int f = 0; f < 5; f++) {
        if (f == TILE_128x64 && P <= 256) continue;  // (two resident 64 x 64 blocks per CU do better there: measured)
        // 128 x 128 (32 KB per MFLOP brought into the CU instead of 48): unpaired products of 1024-token chunks
        if (f == TILE_128x128 && (pair || P < 1024)) continue;
        const long long blocks = (long long)((N + form[f].feat - 1) / form[f].feat) * ((P + form[f].tok - 1) / form[f].tok);
        const double cost = (double)((blocks + cus - 1) / cus) * (form[f].tok * form[f].feat) / form[f].eff;
        if (best < 0 || cost < best_cost * 0.999) {  // ties: the earlier (larger) tile
            best = f;
            best_cost = cost;
        }
    }
    return (TileForm)best;
}

// 1-D grid of the direct-to-LDS tile kernel for ntx feature tiles x nty token tiles (see the kernel's
// block -> tile comment); grids of more than 64 token tiles keep the 2-D form
static dim3 dma_grid(int ntx, int nty, GemmArgs *a)
{
    const unsigned z = a->sk > 1 ? (unsigned)a->sk : 1u;  // split-K family: blockIdx.z = the block's K range
    if (nty > 64) {
        a->ntx = 0; a->nty = 0;
        return dim3(ntx, nty, z);
    }
    a->ntx = ntx; a->nty = nty;
    return dim3((unsigned)((ntx + 7) / 8 * 8 * nty), 1, z);
}

// The epilogue of NV values of ONE lane: feature nb + (lane & 31) (nb: the first of the wave's 32 features in the launch), tokens
// tok[k], stored where on[k].  Everything that depends on the feature alone -- the q | k | v range, the head, the RoPE pair,
// the output column -- is worked out once, the RoPE factors / residuals of all NV values are requested together, then the values
// are finished and stored.  (Until round 6 every value did its own two integer divisions and its own dependent table load
// behind the previous value's store: 16.5 us for the 32 values per lane of the 7B q | k | v launch at 128 tokens.)
// Every lane of the wave calls it together (the RoPE / SwiGLU partner is the adjacent lane's value of the same index).
template <int EPI, int NV>
__device__ __forceinline__ void epi_values(const GemmArgs &a, float (&v)[NV], const int (&tok)[NV], const bool (&on)[NV], int nb, int lane)
{
    int j = nb + (lane & 31), nseg = a.N, seg = 0, ld = a.ldo;
    float *o = a.out;
    if constexpr (EPI == G_QKV) {
        // a wave's 32 columns lie in ONE of the three ranges (launchers: nq and nkv are multiples of 32; a block's tile may
        // lie across two of them -- the stream form's 192-feature tiles)
        seg = nb >= a.nq + a.nkv ? 2 : nb >= a.nq ? 1 : 0;
        j -= seg == 2 ? a.nq + a.nkv : seg == 1 ? a.nq : 0;
        nseg = seg == 0 ? a.nq : a.nkv;
        o = seg == 0 ? a.out : seg == 1 ? a.outk : a.outv;
        ld = seg == 0 ? a.ldo : a.ldkv;
    }
    const bool inr = j < nseg;
    if constexpr (EPI == G_SWIGLU_IL) {
        // row j of the alternating matrix: even W1, odd W3; out[token][j / 2] = silu(a) * b (main.zig:411-416)
        // (sharing a pair's values between its two lanes -- every store with 64 lanes at work -- measured no faster and cost
        // the kernel 80 bytes of scratch: not kept)
#pragma unroll
        for (int k = 0; k < NV; k++) {
            const float partner = __shfl_xor(v[k], 1, 64);
            if (on[k] && !(j & 1) && inr) {
                const float g = swiglu_merge(v[k], partner);
                a.out[(size_t)tok[k] * a.ldo + (j >> 1)] = g;
                if (a.x3_out) planes_store1((__bf16 *)a.x3_out, a.kp_out, tok[k], j >> 1, g);   // the W2 launch's operand, already split
            }
        }
        return;
    }
    constexpr bool kRope = EPI == G_ROPE || EPI == G_ROPE_CACHE || EPI == G_QKV;
    const bool rope = EPI == G_QKV ? seg < 2 : kRope;                                         // (wave-uniform)
    const bool cache = EPI == G_QKV ? seg > 0 : (EPI == G_ROPE_CACHE || EPI == G_CACHE);
    const int hs = a.head_size, jj = inr ? j : 0;
    if constexpr (kRope) {
        if (rope) {   // the pair (j, j + 1) sits in adjacent lanes (main.zig:346-349)
            const float2 *rp = a.rope + ((jj % hs) >> 1);
            float2 cs[NV];
#pragma unroll
            for (int k = 0; k < NV; k++) cs[k] = rp[(size_t)(a.pos0 + (tok[k] < a.P ? tok[k] : 0)) * (size_t)(hs >> 1)];
#pragma unroll
            for (int k = 0; k < NV; k++) {
                const float partner = __shfl_xor(v[k], 1, 64);
                v[k] = (j & 1) ? partner * cs[k].y + v[k] * cs[k].x    // v0*fci + v1*fcr
                               : v[k] * cs[k].x - partner * cs[k].y;   // v0*fcr - v1*fci
            }
        }
    }
    if constexpr (EPI == G_RESID) {   // main.zig:711 (in place too: a lane reads the addresses it writes, nobody else's)
        float rs[NV];
#pragma unroll
        for (int k = 0; k < NV; k++) rs[k] = on[k] && inr ? a.res[(size_t)tok[k] * a.ldres + j] : 0.0f;
#pragma unroll
        for (int k = 0; k < NV; k++) v[k] = rs[k] + v[k];
    }
    if constexpr (EPI == G_SWIGLU) {
        float h[NV];
#pragma unroll
        for (int k = 0; k < NV; k++) h[k] = on[k] && inr ? a.out[(size_t)tok[k] * a.ldo + j] : 0.0f;
#pragma unroll
        for (int k = 0; k < NV; k++) v[k] = swiglu_merge(h[k], v[k]);
    }
    // (token, feature j) -> base + row * pitch: rows of ld floats, or a head-major cache's rows of head_size (kv_index; :354-358)
    size_t base = (size_t)j, pitch = (size_t)ld;
    if (cache && a.kv_head_stride) { base = (size_t)(jj / hs) * a.kv_head_stride + (size_t)(jj % hs); pitch = (size_t)hs; }
#pragma unroll
    for (int k = 0; k < NV; k++)
        if (on[k] && inr) o[base + (size_t)(cache ? a.pos0 + tok[k] : tok[k]) * pitch] = v[k];
}

// epilogue shared by the tile kernels: per MFMA tile a lane owns one feature and 16 tokens
template <int EPI, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs &a, v16f (&acc)[TM][TN], int n0, int m0, int wm,
                                              int wn, int lane, int lo = 0, int hi = 1 << 30)
{
    // [lo, hi): the values i * 16 + r this call finishes (the stream form's ranges share a tile's epilogue; default all)
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int jt = 0; jt < TN; jt++) {
            float v[16];
            int tok[16];
            bool on[16];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                v[r] = acc[i][jt][r];
                tok[r] = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                on[r] = tok[r] < a.P && i * 16 + r >= lo && i * 16 + r < hi;
            }
            epi_values<EPI, 16>(a, v, tok, on, n0 + (wn * TN + jt) * 32, lane);
        }
}

// MFMA 32x32x2 f32: D[i][j] += A[i][k] B[k][j] with
//   A operand: lane l holds A[i = l & 31][k = l >> 5]      -> X[token i][k]
//   B operand: lane l holds B[k = l >> 5][j = l & 31]      -> W[feature j][k]
//   D: lane l, reg r holds D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31]
// (Rounds 1-5 also carried a register-staged form of the tile kernel -- operands through VGPRs and ds_write, any
// K % 4 == 0 -- as the fallback for K % 64 != 0; since round 6 such products run the kernel below over K rounded up to
// whole stages against zero-padded activation rows: pad_k, prefill_common.h.)

// The tile product with the operands brought in by DIRECT-TO-LDS loads
// (global_load_lds_dwordx4, gfx950): no VGPR round trip and no LDS-write instructions -- in the
// register-staged kernel above the copy (6 float4 loads -> 24 ds_write_b32 per thread and stage,
// the row padding forbids wider writes) costs 13 % of the run time (ablation).
//  * LDS tile rows are unpadded (BK = 64 floats = 256 B = 16 float4 slots); a wave-wide load writes
//    1 KB = 4 whole rows, lane L -> row L / 16, physical slot L % 16.
//  * Operands are read with ds_read_b128: lane l of an MFMA 32x32x2 operand holds row l & 31 and
//    k-half l >> 5; it reads the float4 of logical slot 2 s + (l >> 5) of its wave group's 8 slots
//    and feeds components 0..3 to four consecutive MFMAs, i.e. MFMA (s, t) multiplies the k pair
//    (8 s + t, 8 s + 4 + t).  A and B use the same pairing, and a sum over k has no order in exact
//    arithmetic (the fp32 order is fixed by (stage, s, t), then the wave groups: deterministic).
//  * 16 lanes that read the same slot of 16 consecutive rows would hit the same 4 banks (row pitch
//    256 B): the physical slot is  logical ^ (row & 15), applied on the SOURCE address of the
//    load (the LDS side of a direct load is lane-linear), so every 16-lane phase of a b128 read
//    touches 16 different slots.
//  * Two stage buffers; the loads of stage s+1 are issued before stage s is multiplied and waited for
//    (vmcnt(0)) before the barrier that ends it.  Requires K % 64 == 0 (no zero fill on this path);
//    rows past P / N are clamped to the last one and feed outputs nobody stores.
//  * Measured and not kept (7B shape, 512 / 256 tokens, interleaved A/B): three stage buffers with
//    loads two stages ahead (+2.7 % / +7 % time: memory latency is not what the waves wait for, and
//    at 256 tokens the third buffer costs the second resident block); the second wave group issuing
//    its loads mid-stage (+1.4 %); loads and operand reads placed by hand between the MFMA steps
//    with sched_barrier (+9.5 %: hipcc's own interleaving of this loop is better than the pinned one).
//    Operands of super-step s+1 read before the MFMAs of s (two register sets, order pinned with
//    sched_group_barrier): +0.8 % / +3 % -- LDS latency is not it either.
//    PMC of this kernel: SQ_WAIT_ANY 22 % of the wave cycles (parked at the stage barrier), MFMA
//    pipe 66 % busy -- the per-stage barrier with one block of 8 waves per CU is what is left.
//  * PAIR (W1 and W3 of the feed-forward in ONE launch, main.zig:405-416): the block's two n-tiles
//    are the SAME 32 features per wave column of W1 (tile 0) and of W3 (tile 1), so a lane ends up
//    holding both products of its (token, feature) and the epilogue writes silu(a) * b directly --
//    the X tile is loaded once for both, and the [P, hidden] intermediate never goes to memory
//    (the separate W3 launch was the slowest GEMM: it re-read and re-wrote 45 MB at 512 tokens).
//  * Also measured and not kept (512 tokens): stages of half the depth (BK = 32, 48 KB per block,
//    three resident blocks per CU instead of one) +4.9 % time; an XCD-aware block -> tile map that
//    keeps each XCD on one 128-row X tile (L2 resident, only W streams) +0.6 %; neither LDS
//    capacity, nor the memory side, nor latency is what the MFMA pipe waits for.
//  * Also measured and not kept: the non-temporal policy on the W loads (it pays in the short-prompt
//    kernel, where one CU reads a W row once): 512 tokens +-0, 128 tokens +4 % time -- here every W
//    tile is re-read by the blocks of the other token tiles.  3 / 4 stage buffers (counted vmcnt waits, bare s_barrier) for grids
//    that leave every block a CU to itself (row shards, 256-token prompts): +0.8 ... +1.3 % time.  A
//    lone 64 x 64 block already runs at 0.68 of its CU's MFMA peak -- such grids are short of blocks,
//    not of latency hiding.
//  * SPLIT (round 3, the split-K family): grids that leave CUs without a block -- prompts of 33 ... 256 tokens:
//    N = 4096 at 128 tokens is 256 blocks of 32 x 64, one 4-wave block per CU at 0.45 of the peak -- take a
//    LARGER tile and cut K into a.sk contiguous ranges, one block each (blockIdx.z): as many blocks, fewer
//    operand bytes per flop.  A block leaves its accumulators in a.sk_part with write-through stores, drains
//    them and bumps the tile's arrival counter; the block that arrives last -- whichever it is, nobody waits --
//    acquires once, adds the a.sk partials IN RANGE ORDER (its own read back like the others: one fixed
//    order) and runs the epilogue (the hand-off recipe of attention.hip's split kernel).  The sum of range
//    partials rounds differently from one k-ordered chain, so sk is part of the arithmetic: it is chosen from
//    the WHOLE model's shape and the chunk length (choose_sk), never from a rank's share of the rows, and
//    every tile form gives the same bits for the same sk.
//  * SPLIT == 2 (round 3, "k-groups on two blocks"): the two k-groups of a stage -- slots 0..7 and 8..15 of every
//    64-k stage, two wave groups of one block in every other form -- run as TWO blocks of NWG waves each
//    (blockIdx.z = the k-group).  A block brings in only its half of every stage (LDS rows of 8 slots, 128 B;
//    a 1-KB load is 8 half rows; swizzle row & 7: a b128 read's 16-lane phase then meets each bank twice --
//    LDS reads are ~5 % of a stage), so a tile costs half the LDS and twice the blocks exist.  Whichever block
//    of a tile draws its counter first dumps its accumulators (write-through), drains and raises the tile's
//    flag; the other one waits for the flag (the first is running: it drew before), adds the dump to its own
//    accumulators and runs the epilogue.  kg0 + kg1 is exactly the sum the one-block forms form in LDS, and
//    a + b == b + a: THE SAME BITS as every other form of the unsplit family -- so this one is chosen by grid
//    fill alone, shards included.
template <int EPI, int TM, int TN, int KS, bool PAIR = false, int WM = 2, int WN = 2, int SPLIT = 0, bool X3 = false>
__global__ __launch_bounds__(64 * WM * WN * KS) void prefill_gemm_dma(const GemmArgs a)
{
    static_assert(!PAIR || TN == 2, "paired form: one W1 tile and one W3 tile per wave column");
    static_assert(SPLIT != 2 || KS == 2, "k-groups on two blocks: two k-groups");
    constexpr bool KGS = SPLIT == 2;
    constexpr int BK = 64, SLOTS = KGS ? BK / 4 / KS : BK / 4;  // float4 slots per LDS row (KGS: this block's half)
    constexpr int RPI = 64 / SLOTS;                  // tile rows per 1-KB load
    constexpr int BMt = 32 * WM * TM, BNt = 32 * WN * TN;
    constexpr int NWG = WM * WN;                     // waves per k-group: WM x WN of them tile the block
    constexpr int NW = KGS ? NWG : NWG * KS;         // waves
    // X3: the X tile is three planes of bf16 terms, [plane][row][8 slots of 8 bf16] (128 B per row and plane), brought in
    // from the planes matrix a.x3 (launch_split3); a 1-KB load is 8 rows of one plane
    static_assert(!(X3 && KGS), "the planes form has no two-block variant");
    constexpr int XLOADS = X3 ? 3 * BMt / 8 : BMt / RPI;     // 1-KB loads of the X tile per stage
    constexpr int XI = XLOADS / NW, WI = BNt / RPI / NW;      // 1-KB loads per wave and stage
    static_assert(XLOADS % NW == 0 && BNt % (RPI * NW) == 0, "tile rows per wave");
    constexpr int XSTG = X3 ? BMt * 96 : BMt * 4 * SLOTS;     // floats of the X part of a stage
    constexpr int STAGE = XSTG + BNt * 4 * SLOTS;             // floats
    auto swz = [](int row) { return row & (SLOTS - 1); };
    auto swzx = [](int row) { return (row >> 1) & 7; };       // planes: rows r, r + 1 fill one 256-B bank sweep
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wg = KGS ? wave : wave % NWG, kg = KGS ? (int)blockIdx.z : wave / NWG, wm = wg / WN, wn = wg % WN;
    const int kslot0 = KGS ? kg * SLOTS : 0;         // first slot of the 64-k stage this block's LDS rows hold
    // Block -> tile.  2-D grid: x = feature tile, y = token tile.  1-D grid (a.nty > 0, dma_grid): ids are
    // dispatched in order and round-robin over the 8 XCDs, so consecutive groups of 8 nty ids take 8
    // feature tiles x all nty token tiles with id % 8 = feature tile % 8: the nty blocks that read the
    // same W tile run on ONE XCD (one L2 fill) and at the same time -- also in grids of several waves,
    // where the 2-D order lets them drift a whole wave apart.
    int bx = blockIdx.x, by = blockIdx.y;
    if (a.nty > 0) {
        const int group = 8 * a.nty, g = bx / group, local = bx - g * group;
        by = local >> 3;
        bx = g * 8 + (local & 7);
        if (bx >= a.ntx) return;  // padding of the last group
    }
    const int n0 = bx * (PAIR ? BNt / 2 : BNt), m0 = by * BMt;
    // SPLIT == 1: this block's K range (launcher: K % (64 sk) == 0)
    // (whole 64-k stages, as even as they come: range z takes stages [S z / sk, S (z + 1) / sk) of the S = K / 64)
    const int nst = a.K / BK;
    const int kbeg = SPLIT == 1 ? BK * (int)((long long)nst * blockIdx.z / a.sk) : 0;
    const int kend = SPLIT == 1 ? BK * (int)((long long)nst * (blockIdx.z + 1) / a.sk) : a.K;

    // this lane's part of every load: row (within the RPI-row group) lane / SLOTS, physical slot lane % SLOTS
    const int lrow = lane / SLOTS, pslot = lane % SLOTS;
    const float *xsrc[XI], *wsrc[WI];
#pragma unroll
    for (int j = 0; j < XI; j++) {
        if constexpr (X3) {
            const int q = wave * XI + j, plane = q / (BMt / 8), r = (q % (BMt / 8)) * 8 + (lane >> 3), ps = lane & 7;
            xsrc[j] = (const float *)((const __bf16 *)a.x3 + (size_t)min(m0 + r, a.P - 1) * a.ldx3 + (size_t)plane * a.kp +
                                      8 * (ps ^ swzx(r)));
        } else {
            const int r = (wave * XI + j) * RPI + lrow;               // tile row
            xsrc[j] = a.x + (size_t)min(m0 + r, a.P - 1) * a.ldx + 4 * (kslot0 + (pslot ^ swz(r)));
        }
    }
#pragma unroll
    for (int j = 0; j < WI; j++) {
        const int r = (wave * WI + j) * RPI + lrow;
        if (PAIR) {  // LDS row r = wave column r / 64, tile (r % 64) / 32 (0: W1, 1: W3), feature r % 32
            const float *m = ((r & 63) >> 5) ? a.w2 : a.w;
            const int f = n0 + (r >> 6) * 32 + (r & 31);
            wsrc[j] = m + (size_t)min(f, a.N - 1) * a.ldw + 4 * (kslot0 + (pslot ^ swz(r)));
        } else if (EPI == G_QKV) {  // the matrix of the block's column range
            const int seg = n0 >= a.nq + a.nkv ? 2 : n0 >= a.nq ? 1 : 0;
            const float *m = seg == 0 ? a.w : seg == 1 ? a.wk : a.wv;
            const int f0 = n0 - (seg == 2 ? a.nq + a.nkv : seg == 1 ? a.nq : 0), nseg = seg == 0 ? a.nq : a.nkv;
            wsrc[j] = m + (size_t)min(f0 + r, nseg - 1) * a.ldw + 4 * (kslot0 + (pslot ^ swz(r)));
        } else {
            wsrc[j] = a.w + (size_t)min(n0 + r, a.N - 1) * a.ldw + 4 * (kslot0 + (pslot ^ swz(r)));
        }
    }
#define L2Z_DMA_ISSUE(k0_, buf_)                                                                          \
    do {                                                                                                  \
        float *xs_ = smem + (buf_) * STAGE, *ws_ = xs_ + XSTG;                                            \
        _Pragma("unroll") for (int j = 0; j < XI; j++)   /* (planes: k0 bf16 = k0 / 2 floats) */          \
            lds_dma16(xsrc[j] + (X3 ? (k0_) / 2 : (k0_)), xs_ + (wave * XI + j) * 256);  /* a 1-KB load: RPI whole LDS rows */ \
        _Pragma("unroll") for (int j = 0; j < WI; j++)                                                    \
            lds_dma16(wsrc[j] + (k0_), ws_ + (wave * WI + j) * 256);                                       \
    } while (0)

    v16f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    // operand addresses (float4 units) of super-step s: row base + ((logical slot) ^ (row & 15))
    const int hl = lane >> 5, il = lane & 31;
    constexpr int SPG = KGS ? SLOTS : SLOTS / KS;   // slots of this wave group per stage
    constexpr int SS = SPG / 2;       // super-steps (one float4 per k-half each)
    int arow[TM], brow[TN], asw[TM], bsw[TN];
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int r = wm * 32 * TM + i * 32 + il;
        arow[i] = X3 ? r * 8 : r * SLOTS;
        asw[i] = X3 ? swzx(r) : swz(r);
    }
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int r = wn * 32 * TN + j * 32 + il;
        brow[j] = r * SLOTS;
        bsw[j] = swz(r);
    }

#define L2Z_MULTIPLY_STAGE(buf_)                                                                         \
    do {                                                                                                  \
        const v4f *xr = (const v4f *)(smem + (buf_) * STAGE), *wr = xr + XSTG / 4;                         \
        _Pragma("unroll") for (int s = 0; s < SS; s++) {                                                   \
            const int slot = (KGS ? 0 : kg * SPG) + 2 * s + hl;                                           \
            v4f av[TM], bv[TN];                                                                           \
            _Pragma("unroll") for (int i = 0; i < TM; i++) av[i] = xr[arow[i] + (slot ^ asw[i])];          \
            _Pragma("unroll") for (int j = 0; j < TN; j++) bv[j] = wr[brow[j] + (slot ^ bsw[j])];          \
            /* four MFMAs back to back on ONE accumulator, then the next accumulator: a dependent MFMA */ \
            /* issues at full rate only straight behind its producer (anything in between, even an    */ \
            /* MFMA on another accumulator, costs tens of cycles per step: MI355X_MICROARCH.md)        */ \
            _Pragma("unroll") for (int i = 0; i < TM; i++)                                                 \
                _Pragma("unroll") for (int j = 0; j < TN; j++)                                             \
                    _Pragma("unroll") for (int t = 0; t < 4; t++)                                          \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][t], bv[j][t], acc[i][j], 0, 0, 0); \
        }                                                                                                 \
    } while (0)

    // X3: the same product on the bf16 matrix cores (prefill_common.h split3 / x3_mfma).  A k-step is 16 k: lane (row,
    // k-half hl) takes the row's 8 consecutive k of bf16 slot sa = (8 / KS) kg + 2 u + hl -- X: one b128 read per plane, the
    // terms are already split (launch_split3); W: the two float4 of slots 2 sa, 2 sa + 1, split here -- and feeds the six
    // products to ONE accumulator.
    // The loop is software-pipelined by hand: one block of <= 8 waves owns the CU, so nothing overlaps unless a wave
    // overlaps with itself.  While the MFMAs of k-step t run, the wave (a) reads W's floats of step t + 1 and splits them
    // between the MFMAs, (b) refills each token tile's X operand for t + 1 right behind the MFMAs that used it, (c) in the
    // last step of a stage, behind the barrier that says "stage s + 1 has landed and nobody reads stage s any more",
    // issues the loads of stage s + 2, spread over the step.  One barrier per stage; loads run a whole stage ahead.
    if constexpr (X3) {
        constexpr int NU = 4 / KS;                   // k-steps of this wave's k-group per stage
        constexpr int NL = XI + WI;                  // this wave's 1-KB loads per stage
        const int nstage = (kend - kbeg) / BK;
        auto issue_one = [&](int k0_, int buf_, int idx) {
            float *xs_ = smem + buf_ * STAGE, *ws_ = xs_ + XSTG;
            if (idx < XI) lds_dma16(xsrc[idx] + k0_ / 2, xs_ + (wave * XI + idx) * 256);
            else lds_dma16(wsrc[idx - XI] + k0_, ws_ + (wave * WI + (idx - XI)) * 256);
        };
        auto read_a = [&](int buf_, int u, int i) {
            const v8bf *xp = (const v8bf *)(smem + buf_ * STAGE);
            const int o = arow[i] + ((kg * (8 / KS) + 2 * u + hl) ^ asw[i]);
            Bf3 r;
            r.t1 = xp[o]; r.t2 = xp[BMt * 8 + o]; r.t3 = xp[2 * BMt * 8 + o];
            return r;
        };
        auto read_b = [&](int buf_, int u, int j, v4f &lo, v4f &hi) {
            const v4f *wr = (const v4f *)(smem + buf_ * STAGE + XSTG);
            const int sb = 2 * (kg * (8 / KS) + 2 * u + hl);
            lo = wr[brow[j] + (sb ^ bsw[j])];
            hi = wr[brow[j] + ((sb + 1) ^ bsw[j])];
        };
        Bf3 av[TM], bcur[TN];
        v4f blo[TN], bhi[TN];
#pragma unroll
        for (int q = 0; q < NL; q++) issue_one(kbeg, 0, q);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (nstage > 1) {
#pragma unroll
            for (int q = 0; q < NL; q++) issue_one(kbeg + BK, 1, q);
        }
#pragma unroll
        for (int j = 0; j < TN; j++) { read_b(0, 0, j, blo[j], bhi[j]); bcur[j] = split3(blo[j], bhi[j]); }
#pragma unroll
        for (int i = 0; i < TM; i++) av[i] = read_a(0, 0, i);
        int buf = 0;
        // one k-step; the flags are compile-time so that the steady-state loop body is ONE basic block
        auto step = [&](auto last_u_c, auto has_next_c, auto issue_c, int s, int u) {
            constexpr bool last_u = decltype(last_u_c)::value, has_next = decltype(has_next_c)::value,
                           issue = decltype(issue_c)::value;
            const int nb = last_u ? buf ^ 1 : buf, nu = last_u ? 0 : u + 1;
            if constexpr (last_u && has_next) {
                // stage s + 1 has landed (its loads were issued a stage ago) and this wave's reads of stage s are done ...
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                if (!(L2Z_X3_EXP & 16))
                __syncthreads();   // ... for every wave: stage s's buffer is free for the loads of stage s + 2
            }
            if constexpr (has_next && !(L2Z_X3_EXP & 8)) {
#pragma unroll
                for (int j = 0; j < TN; j++) read_b(nb, nu, j, blo[j], bhi[j]);
            }
            Bf3 bnxt[TN];
#pragma unroll
            for (int i = 0; i < TM; i++) {
#pragma unroll
                for (int j = 0; j < TN; j++)
                    if (!(L2Z_X3_EXP & 2)) acc[i][j] = x3_mfma(av[i], bcur[j], acc[i][j]);
                if constexpr (issue && !(L2Z_X3_EXP & 1)) {
                    constexpr int per = (NL + TM - 1) / TM;
#pragma unroll
                    for (int q = i * per; q < (i + 1) * per && q < NL; q++) issue_one(kbeg + (s + 2) * BK, buf, q);
                }
                if constexpr (has_next) {
                    if (!(L2Z_X3_EXP & 4)) av[i] = read_a(nb, nu, i);
                    // W's terms for the next step: tile j is split behind token tile (j + 1) % TM's MFMAs (its floats
                    // were requested at the top of the step)
#pragma unroll
                    for (int j = 0; j < TN; j++)
                        if ((j + 1) % TM == i) bnxt[j] = (L2Z_X3_EXP & 8) ? bcur[j] : split3(blo[j], bhi[j]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (has_next) {
#pragma unroll
                for (int j = 0; j < TN; j++) bcur[j] = bnxt[j];
            }
        };
        using T = std::true_type;
        using F = std::false_type;
        int s = 0;
        for (; s + 2 < nstage; s++) {   // steady state: a next step always exists, the last step of a stage issues loads
#pragma unroll
            for (int u = 0; u < NU; u++) {
                if (u == NU - 1) step(T{}, T{}, T{}, s, u);
                else step(F{}, T{}, F{}, s, u);
            }
            buf ^= 1;
        }
        if (s + 1 < nstage) {           // the stage before the last: nothing left to load
#pragma unroll
            for (int u = 0; u < NU; u++) {
                if (u == NU - 1) step(T{}, T{}, F{}, s, u);
                else step(F{}, T{}, F{}, s, u);
            }
            buf ^= 1;
            s++;
        }
#pragma unroll
        for (int u = 0; u < NU; u++) {  // the last stage: its last step has no successor
            if (u == NU - 1) step(T{}, F{}, F{}, s, u);
            else step(F{}, T{}, F{}, s, u);
        }
        __syncthreads();   // every wave is done with the stage buffers (the k-groups' sums reuse them)
    } else {
    L2Z_DMA_ISSUE(kbeg, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        if (k0 + BK < kend) L2Z_DMA_ISSUE(k0 + BK, buf ^ 1);
        L2Z_MULTIPLY_STAGE(buf);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's loads of the next stage have landed
        __syncthreads();                                  // everyone's have, and nobody still reads this one
        buf ^= 1;
    }
    }
#undef L2Z_MULTIPLY_STAGE
    if (KS > 1 && !KGS) {
        float *red = smem;  // [KS-1][NWG waves][TM*TN*16][64 lanes]
        if (kg > 0) {
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        red[((((kg - 1) * NWG + wg) * TM * TN + i * TN + j) * 16 + r) * 64 + lane] = acc[i][j][r];
        }
        __syncthreads();
        if (kg > 0) return;
#pragma unroll
        for (int g = 1; g < KS; g++)
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        acc[i][j][r] += red[((((g - 1) * NWG + wave) * TM * TN + i * TN + j) * 16 + r) * 64 + lane];
    }
#undef L2Z_DMA_ISSUE
    if constexpr (SPLIT == 2) {
        // every wave of this block holds its k-group's sums for its 32 TM x 32 TN outputs
        constexpr int PT = NWG * TM * TN * 16 * 64;  // floats per dump = the tile's outputs
        const int ntx_all = a.nty > 0 ? a.ntx : (int)gridDim.x;
        const size_t tile = (size_t)by * ntx_all + bx;
        float *dump = a.sk_part + tile * PT + (size_t)wg * (TM * TN * 16 * 64) + lane;
        int *cnt = a.sk_cnt + 2 * tile, *flag = cnt + 1;
        int *role = (int *)smem;
        __syncthreads();  // every wave is done with the stage buffers
        if (tid == 0) *role = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*role == 0) {  // drew first: leave the sums for the other block of this tile
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        __hip_atomic_store(dump + ((i * TN + j) * 16 + r) * 64, acc[i][j][r], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);  // write-through
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its part
            __syncthreads();
            if (tid == 0) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (tid == 0) {  // drew second: the first one is running (it drew before) and will raise the flag
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(2);
            __hip_atomic_store(flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch
            __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // this CU's stale lines of the dump
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] += dump[((i * TN + j) * 16 + r) * 64];  // kg0 + kg1 (a + b == b + a)
    }
    if constexpr (SPLIT == 1) {
        // the k-group-0 waves hold the block's sums (the others have left; a barrier only counts live waves)
        constexpr int PT = NWG * TM * TN * 16 * 64;  // floats per partial = the tile's outputs
        const int ntx_all = a.nty > 0 ? a.ntx : (int)gridDim.x;
        const size_t tile = (size_t)by * ntx_all + bx;
        float *part = a.sk_part + tile * (size_t)a.sk * PT;
        float *mine = part + (size_t)blockIdx.z * PT + (size_t)wg * (TM * TN * 16 * 64) + lane;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    __hip_atomic_store(mine + ((i * TN + j) * 16 + r) * 64, acc[i][j][r], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);  // write-through
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its partial
        __syncthreads();
        int *flag = (int *)smem;  // stage buffers and the k-group sums are dead
        if (tid == 0) {
            const int prev = __hip_atomic_fetch_add(a.sk_cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = prev == a.sk - 1;
            if (last) {
                __hip_atomic_store(a.sk_cnt + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // this CU's stale lines of the partials
            }
            *flag = last;
        }
        __syncthreads();
        if (!*flag) return;
        const float *p0 = part + (size_t)wg * (TM * TN * 16 * 64) + lane;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    float v = p0[((i * TN + j) * 16 + r) * 64];  // range 0, then 1, ... in order
                    for (int z = 1; z < a.sk; z++) v += p0[(size_t)z * PT + ((i * TN + j) * 16 + r) * 64];
                    acc[i][j][r] = v;
                }
    }
    if constexpr (PAIR) {
        const int j = n0 + wn * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int tok = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (tok < a.P && j < a.N) {
                    const float g = swiglu_merge(acc[i][0][r], acc[i][1][r]);  // :411-416
                    a.out[(size_t)tok * a.ldo + j] = g;
                    if (a.x3_out) planes_store1((__bf16 *)a.x3_out, a.kp_out, tok, j, g);   // the W2 launch's operand, already split
                }
            }
    } else {
        gemm_epilogue<EPI, TM, TN>(a, acc, n0, m0, wm, wn, lane);
    }
}

// ---- the STREAM form of the planes kernel: chunks of <= 128 tokens of matrices that stream from HBM (round 6) ----
// At <= 128 tokens a layer's matrices cross the chip once (809 MB at the 7B shape: 130 us at the HBM rate) against 40-160 us
// of bf16 MFMAs: the tile forms above -- one or two blocks per CU, two stage buffers, loads ONE stage ahead -- are bound by
// neither: every stage waits out the memory latency.  This form keeps more of the stream in flight and splits W once:
//  * a block = ALL the chunk's tokens (TM tiles of 32) x 128 features x one K range; 8 waves = 4 feature groups x 2
//    k-groups; a wave owns 32 features x all tokens, so a W element is split into its bf16 terms exactly once on the chip;
//  * stages of 32 k (X planes 192 B per token, W 128 B per feature: 22-40 KB) in a ring of NBUF = 4-7 buffers, loads
//    NBUF stages ahead: the stage a wave needs next was requested 3-6 stages ago (s_waitcnt vmcnt counted, one barrier per
//    stage); the loop is the hand-pipelined one of the tile forms (operands of stage s + 1 read / split behind stage s's MFMAs);
//  * the grid is filled by K ranges (blockIdx.z; x3_stream_sk: a function of the WHOLE model's matrix and the chunk length),
//    partial sums through the split-K workspace, the last arriver adds them in range order and runs the epilogue.
// An output's value is a function of (K, the ranges) only -- not of the rows a rank owns, not of launch fusion.
#ifdef L2Z_X3_TIMELINE
// measurement build (scripts/x3_timeline.sh): wall-clock stamps (100 MHz) of every block of the LAST launch of each epilogue
// kind: [0] entry, [1] first stage landed, [2] loop done, [3] k-groups summed, [4] partial sums drained, [5] siblings arrived,
// [6] the ranges' sums read and added, [7] end
constexpr int kXtlBlocks = 1024;
__device__ long long g_xtl[8 * kXtlBlocks * 8];
#define L2Z_XTL(i) tl[i] = wall_clock64()
#else
#define L2Z_XTL(i) do { } while (0)
#endif

template <int N_> __device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N_) : "memory");
}

template <int EPI, int TM, int NBUF, int WN>
__global__ __launch_bounds__(128 * WN) void prefill_x3_stream(const GemmArgs a)
{
    constexpr int BK = 32, KS = 2, NW = WN * KS, BMt = 32 * TM, BNt = 32 * WN, TN = 1;
    constexpr int XLOADS = 3 * BMt * 64 / 1024;              // 1-KB loads of a stage's X planes: 16 rows of one plane each
    // per wave and stage (X: the last waves repeat a load); experiment builds 32 / 64: no X / no W loads at all
    constexpr int XI = (L2Z_X3_EXP & 32) ? 0 : (XLOADS + NW - 1) / NW, WI = (L2Z_X3_EXP & 64) ? 0 : BNt * 128 / 1024 / NW, NL = XI + WI;
    constexpr int XSTG = 3 * BMt * 16, WSTG = BNt * 32, STAGE = XSTG + WSTG;                 // floats
    static_assert(NL * (NBUF - 1) <= 63, "vmcnt");
    extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef L2Z_X3_TIMELINE
    long long tl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto tl_store = [&]() {
        tl[7] = wall_clock64();
        if (threadIdx.x == 0 && blockIdx.x < kXtlBlocks)
            for (int i = 0; i < 8; i++) g_xtl[((size_t)EPI * kXtlBlocks + blockIdx.x) * 8 + i] = tl[i];
    };
#else
    auto tl_store = []() {};
#endif
    L2Z_XTL(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wn = wave % WN, kg = wave / WN;
    const int hl = lane >> 5, il = lane & 31;
    // block -> (tile, range): the ranges of a tile are CONSECUTIVE block ids -- dispatched together, finishing together,
    // so that they can share the tile's reduction and epilogue (below) without anybody waiting for a block not yet resident
    const int nst_all = a.K / BK, sk = a.sk > 1 ? a.sk : 1;
    // (a.ntx > 0: the launch does not fit the chip in one round -- range-major ids, the last arriver of a tile finishes it)
    const bool coop = a.ntx == 0;
    const int bx = coop ? (int)blockIdx.x / sk : (int)blockIdx.x % a.ntx, bz = coop ? (int)blockIdx.x - bx * sk : (int)blockIdx.x / a.ntx;
    const int n0 = bx * BNt;
    const int sbeg = (int)((long long)nst_all * bz / sk), nstage = (int)((long long)nst_all * (bz + 1) / sk) - sbeg;
    const int kbeg = sbeg * BK;
    auto swzx = [](int r) { return (r >> 2) & 3; };          // plane rows of 64 B: four rows fill one 256-B bank sweep
    auto swzw = [](int r) { return (r >> 1) & 7; };          // W rows of 128 B: two rows
    const float *xsrc[XI > 0 ? XI : 1], *wsrc[WI > 0 ? WI : 1];
    int xdst[XI > 0 ? XI : 1];
#pragma unroll
    for (int j = 0; j < XI; j++) {
        const int q = min(wave * XI + j, XLOADS - 1), plane = q / (2 * TM), r = (q % (2 * TM)) * 16 + (lane >> 2), ps = lane & 3;
        xsrc[j] = (const float *)((const __bf16 *)a.x3 + (size_t)min(r, a.P - 1) * a.ldx3 + (size_t)plane * a.kp + 8 * (ps ^ swzx(r)));
        xdst[j] = q * 256;
    }
#pragma unroll
    for (int j = 0; j < WI; j++) {
        const int r = (wave * WI + j) * 8 + (lane >> 3), ps = lane & 7;
        const float *m = a.w;
        int f = n0 + r, nseg = a.N;
        if constexpr (EPI == G_QKV) {
            const int seg = f >= a.nq + a.nkv ? 2 : f >= a.nq ? 1 : 0;   // (per row: a tile may lie across two of the matrices)
            m = seg == 0 ? a.w : seg == 1 ? a.wk : a.wv;
            f -= seg == 2 ? a.nq + a.nkv : seg == 1 ? a.nq : 0;
            nseg = seg == 0 ? a.nq : a.nkv;
        }
        wsrc[j] = m + (size_t)min(f, nseg - 1) * a.ldw + 4 * (ps ^ swzw(r));
    }
    auto issue_one = [&](int stage, int idx) {
        float *xs_ = smem + (stage % NBUF) * STAGE, *ws_ = xs_ + XSTG;
        const int k0 = kbeg + stage * BK;
        // (W: non-temporal -- one CU reads a weight once per launch; a load-only walk of this mix streams 6.9 instead of
        // 6.05 TB/s that way: scripts/lds_fill_probe.hip.  The planes: every block of a K range re-reads them from the L2.)
        if (idx < XI) lds_dma16(xsrc[idx] + k0 / 2, xs_ + xdst[idx]);
        else if (L2Z_X3_EXP & 128) lds_dma16(wsrc[idx - XI] + k0, ws_ + (wave * WI + (idx - XI)) * 256);
        else lds_dma16_nt(wsrc[idx - XI] + k0, ws_ + (wave * WI + (idx - XI)) * 256);
    };
    int arow[TM], asw[TM];
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int r = i * 32 + il;
        arow[i] = r * 4;
        asw[i] = swzx(r);
    }
    const int brow = (wn * 32 + il) * 8, bsw = swzw(wn * 32 + il);
    const int sa = kg * 2 + hl, sb = 2 * sa;                 // this lane's bf16 slot / first float4 slot of a stage row
    auto read_a = [&](int stage, int i) {
        const v8bf *xp = (const v8bf *)(smem + (stage % NBUF) * STAGE);
        const int o = arow[i] + (sa ^ asw[i]);
        Bf3 r;
        r.t1 = xp[o]; r.t2 = xp[BMt * 4 + o]; r.t3 = xp[2 * BMt * 4 + o];
        return r;
    };
    auto read_b = [&](int stage, v4f &lo, v4f &hi) {
        const v4f *wr = (const v4f *)(smem + (stage % NBUF) * STAGE + XSTG);
        lo = wr[brow + (sb ^ bsw)];
        hi = wr[brow + ((sb + 1) ^ bsw)];
    };
    v16f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][0][r] = 0.0f;

    // prologue: the first NBUF stages requested; stage 0 awaited
#pragma unroll
    for (int b = 0; b < NBUF; b++)
        if (b < nstage) {
#pragma unroll
            for (int q = 0; q < NL; q++) issue_one(b, q);
        }
    {
        const int ahead = (nstage < NBUF ? nstage : NBUF) - 1;   // stages that may still be in flight
        // (a literal per case: the counter takes an immediate)
        if (ahead <= 0) wait_vmcnt<0>();
        else if (ahead == 1) wait_vmcnt<NL>();
        else if (ahead == 2) wait_vmcnt<2 * NL>();
        else if (ahead == 3) wait_vmcnt<(NBUF > 3 ? 3 : 0) * NL>();
        else if (ahead == 4) wait_vmcnt<(NBUF > 4 ? 4 : 0) * NL>();
        else if (ahead == 5) wait_vmcnt<(NBUF > 5 ? 5 : 0) * NL>();
        else wait_vmcnt<(NBUF > 6 ? 6 : 0) * NL>();
    }
    __syncthreads();
    L2Z_XTL(1);
    Bf3 av[TM], bcur;
    v4f blo, bhi;
    read_b(0, blo, bhi);
    bcur = split3(blo, bhi);
#pragma unroll
    for (int i = 0; i < TM; i++) av[i] = read_a(0, i);

    // step s: MFMAs of stage s (operands in registers); behind them the loads of stage s + NBUF and the operands of s + 1.
    // AHEAD: stages beyond s + 1 that may still be in flight at the top of the step (steady state NBUF - 2)
    auto step = [&](auto ahead_c, auto has_next_c, auto issue_c, int s) {
        constexpr int ahead = decltype(ahead_c)::value;
        constexpr bool has_next = decltype(has_next_c)::value, issue = decltype(issue_c)::value;
        if constexpr (has_next) {
            wait_vmcnt<ahead * NL>();   // stage s + 1 has landed (this wave's part) and this wave's reads of stage s are done ...
            // ... for every wave: stage s's buffer takes the loads of stage s + NBUF.  (The bare barrier: behind __syncthreads()
            // the compiler drains vmcnt to 0 in the steps that issue nothing -- the tail -- and the last stages then land with
            // nothing multiplying beside them.)
            if (!(L2Z_X3_EXP & 16)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (!(L2Z_X3_EXP & 8)) read_b(s + 1, blo, bhi);
        }
        Bf3 bnxt;
#pragma unroll
        for (int i = 0; i < TM; i++) {
            if (!(L2Z_X3_EXP & 2)) acc[i][0] = x3_mfma(av[i], bcur, acc[i][0]);
            if constexpr (issue && !(L2Z_X3_EXP & 1)) {
                constexpr int per = (NL + TM - 1) / TM;
#pragma unroll
                for (int q = i * per; q < (i + 1) * per && q < NL; q++) issue_one(s + NBUF, q);
            }
            if constexpr (has_next) {
                if (!(L2Z_X3_EXP & 4)) av[i] = read_a(s + 1, i);
                if (i == (TM > 1 ? 1 : 0)) bnxt = (L2Z_X3_EXP & 8) ? bcur : split3(blo, bhi);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (has_next) bcur = bnxt;
    };
    using T = std::true_type;
    using F = std::false_type;
    int s = 0;
    for (; s + NBUF < nstage; s++) step(std::integral_constant<int, NBUF - 2>{}, T{}, T{}, s);
    // the tail: nothing left to request; fewer and fewer stages in flight
#pragma unroll
    for (int t = NBUF - 1; t >= 1; t--)
        if (nstage - 1 - s == t) {
            // stages s + 1 .. s + t are the last ones: t - 1 of them beyond s + 1
            if (t - 1 >= NBUF - 2) step(std::integral_constant<int, NBUF - 2>{}, T{}, F{}, s);
            else if (t - 1 == 4) step(std::integral_constant<int, (NBUF > 6 ? 4 : 0)>{}, T{}, F{}, s);
            else if (t - 1 == 3) step(std::integral_constant<int, (NBUF > 5 ? 3 : 0)>{}, T{}, F{}, s);
            else if (t - 1 == 2) step(std::integral_constant<int, (NBUF > 4 ? 2 : 0)>{}, T{}, F{}, s);
            else if (t - 1 == 1) step(std::integral_constant<int, (NBUF > 3 ? 1 : 0)>{}, T{}, F{}, s);
            else step(std::integral_constant<int, 0>{}, T{}, F{}, s);
            s++;
        }
    step(std::integral_constant<int, 0>{}, F{}, F{}, s);
    __syncthreads();   // every wave is done with the ring (the k-groups' sums reuse it)
    L2Z_XTL(2);

    // the two k-groups' sums, then (sk > 1) the ranges' through the workspace, then the epilogue -- as in the tile forms
    {
        float *red = smem;  // [WN waves][TM * 16][64 lanes]
        if (kg > 0) {
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) red[((wn * TM + i) * 16 + r) * 64 + lane] = acc[i][0][r];
        }
        __syncthreads();
        if (kg > 0) return;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][0][r] += red[((wn * TM + i) * 16 + r) * 64 + lane];
    }
    L2Z_XTL(3);
    if constexpr (EPI == G_RESID) {
        if (sk > 1 && a.defer) {
            // the sums stay as they are: the next rmsnorm launch adds the ranges in order and the residual (DeferredSum)
            float *mine = a.sk_part + ((size_t)bx * (size_t)sk + (size_t)bz) * (WN * TM * 16 * 64) + (size_t)wn * (TM * 16 * 64) + lane;
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) mine[(i * 16 + r) * 64] = acc[i][0][r];
            tl_store();
            return;
        }
    }
    if (sk > 1 && !coop) {
        // A launch of several rounds of blocks: whichever block of a tile arrives last adds the sk partials in range order
        // (nobody waits: a waiting block would hold a CU its not-yet-resident siblings need) and runs the epilogue.
        constexpr int PT = WN * TM * 16 * 64;   // floats per partial = the tile's outputs
        float *part = a.sk_part + (size_t)bx * (size_t)sk * PT;
        float *mine = part + (size_t)bz * PT + (size_t)wn * (TM * 16 * 64) + lane;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                __hip_atomic_store(mine + (i * 16 + r) * 64, acc[i][0][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // (the live waves: k-group 0)
        int *flag = (int *)smem + WN * TM * 16 * 64;   // past the k-groups' sums
        if (tid == 0) {
            const int prev = __hip_atomic_fetch_add(a.sk_cnt + 2 * bx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = prev == sk - 1;
            if (last) __hip_atomic_store(a.sk_cnt + 2 * bx, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch
            *flag = last;
        }
        __syncthreads();
        if (!*flag) return;
        // range 0, then 1, ... in order; a range's TM x 16 values are requested together (device-scope loads: served past this
        // XCD's caches)
        const float *p0 = part + (size_t)wn * (TM * 16 * 64) + lane;
        auto fetch = [&](int z, v16f (&d)[TM]) {
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    d[i][r] = __hip_atomic_load(p0 + (size_t)z * PT + (i * 16 + r) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        {
            v16f t[TM];
            fetch(0, t);
#pragma unroll
            for (int i = 0; i < TM; i++) acc[i][0] = t[i];
            for (int z = 1; z < sk; z++) {   // (one round trip per range; two ranges in flight cost 64 more registers and spilled)
                fetch(z, t);
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[i][0][r] += t[i][r];
            }
        }
    } else if (sk > 1) {
        // The tile's K ranges: every block leaves its sums in the workspace (write-through), arrives, and waits for its sk - 1
        // siblings (consecutive block ids: resident with it or about to be -- see the block -> tile map); then EVERY block
        // finishes 1 / sk of the tile: values [lo, lo + PER) of each lane's TM x 16 (flat index i * 16 + r), the ranges added
        // IN RANGE ORDER -- a range's PER values requested together, one round trip per range -- and their epilogue.
        // (The last-arriver form: one block re-read all sk partials of the tile with its 7 siblings' CUs idle.)
        constexpr int PT = WN * TM * 16 * 64;   // floats per partial = the tile's outputs
        float *part = a.sk_part + (size_t)bx * (size_t)sk * PT;
        float *mine = part + (size_t)bz * PT + (size_t)wn * (TM * 16 * 64) + lane;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                __hip_atomic_store(mine + (i * 16 + r) * 64, acc[i][0][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // (the live waves: k-group 0) every wave's part has left
        L2Z_XTL(4);
        int *arrive = a.sk_cnt + 2 * bx, *done = arrive + 1;
        if (tid == 0) {
            __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the partials were written through and drained)
            // (bounded: ~1 s; a sibling that never arrives -- a launch that failed half way -- must not hang the device)
            for (int it = 0; __hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < sk && it < (1 << 24); it++)
                __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        L2Z_XTL(5);
        const float *p0 = part + (size_t)wn * (TM * 16 * 64) + lane;
        auto finish = [&](auto sk_c) {
            constexpr int SK = decltype(sk_c)::value, PER = TM * 16 / SK;   // values per block
            const int lo = bz * PER;
            // ALL the ranges' values of this block are requested together (SK x PER = TM x 16 loads, one round trip), then
            // added in range order.  Device-scope loads: served past this XCD's caches (an acquire fence per block would
            // drop the whole L2).
            float t[SK][PER];
#pragma unroll
            for (int z = 0; z < SK; z++)
#pragma unroll
                for (int k = 0; k < PER; k++)
                    t[z][k] = __hip_atomic_load(p0 + (size_t)z * PT + (size_t)(lo + k) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int z = 1; z < SK; z++)
#pragma unroll
                for (int k = 0; k < PER; k++) t[0][k] += t[z][k];
            if (tid == 0) {   // the last block to have read leaves the counters at zero for the next launch
                const int prev = __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (prev == sk - 1) {
                    __hip_atomic_store(arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            L2Z_XTL(6);
            int tok[PER];
            bool on[PER];
#pragma unroll
            for (int k = 0; k < PER; k++) {
                const int idx = lo + k, i = idx >> 4, r = idx & 15;
                tok[k] = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                on[k] = tok[k] < a.P;
            }
            epi_values<EPI, PER>(a, t[0], tok, on, n0 + wn * 32, lane);
        };
        // (sk is 2, 4 or 8: x3_stream_sk; TM * 16 divides)
        if (sk == 2) finish(std::integral_constant<int, 2>{});
        else if (sk == 4) finish(std::integral_constant<int, 4>{});
        else finish(std::integral_constant<int, 8>{});
        tl_store();
        return;
    }
    gemm_epilogue<EPI, TM, TN>(a, acc, n0, 0, 0, wn, lane);
    tl_store();
}

// A form of the tile kernel as a launch: kernel, threads, dynamic LDS.  The block's output tile is 32 WM TM tokens x
// 32 WN TN features whichever cores multiply it: on the f32 matrix cores WM x WN waves per k-group tile it; on the bf16
// ones (x3: the launch's activations are planes of bf16 terms, a.x3) WN waves each take ALL the tile's tokens x 32 TN features, so that a W element is split
// into its bf16 terms by one wave only -- the same template with (TM WM, TN, 1, WN).  Same k order either way inside a
// mode; between the modes the arithmetic differs (tolerance, not bits).
struct DmaForm { const void *fn; unsigned threads; size_t lds; };
template <int EPI, int TM, int TN, int KS, bool PAIR = false, int WM = 2, int WN = 2, int SPLIT = 0>
DmaForm dma_form(bool x3)
{
    constexpr size_t BMt = 32 * WM * TM, BNt = 32 * WN * TN;
    const size_t red = SPLIT == 2 ? 0 : (size_t)(KS - 1) * WM * WN * TM * TN * 16 * 64 * sizeof(float);  // the k-groups' sums
    DmaForm f;
    if constexpr (SPLIT != 2) {
        if (x3) {
            f.fn = (const void *)prefill_gemm_dma<EPI, TM * WM, TN, KS, PAIR, 1, WN, SPLIT, true>;
            f.threads = 64 * WN * KS;
            f.lds = 2 * (BMt * 384 + BNt * 256);   // two stages of three 128-B plane rows per token + 256-B W rows
            if (red > f.lds) f.lds = red;
            if (f.lds > 48 * 1024) (void)hipFuncSetAttribute(f.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.lds);
            return f;
        }
    }
    f.fn = (const void *)prefill_gemm_dma<EPI, TM, TN, KS, PAIR, WM, WN, SPLIT, false>;
    f.threads = SPLIT == 2 ? 64 * WM * WN : 64 * WM * WN * KS;
    f.lds = 2 * (BMt + BNt) * (SPLIT == 2 ? 32 : 64) * sizeof(float);   // two stage buffers (SPLIT == 2: of half rows)
    if (red > f.lds) f.lds = red;
    if (f.lds > 48 * 1024) (void)hipFuncSetAttribute(f.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.lds);
    return f;
}

template <int EPI, int TM, int TN, int KS>
hipError_t gemm_launch_t(const GemmArgs &a, hipStream_t st)
{
    constexpr int BMt = 64 * TM, BNt = 64 * TN;
    static_assert((BMt / 4) % (4 * KS) == 0 && (BNt / 4) % (4 * KS) == 0, "tile rows per wave");
    if (a.K % 64 != 0 || a.ldx % 4 != 0) return hipErrorInvalidValue;   // (the launchers round K up: pad_k)
    const DmaForm f = dma_form<EPI, TM, TN, KS, false>(a.x3 != nullptr);
    GemmArgs args = a;
    const dim3 grid1 = dma_grid((a.N + BNt - 1) / BNt, (a.P + BMt - 1) / BMt, &args);
    void *params[] = {&args};
    return hipLaunchKernel(f.fn, grid1, dim3(f.threads), params, f.lds, st);
}

// the direct-to-LDS tile kernel with fewer waves per block -- 32 x 64 (1 x 2 waves per k-group) and
// 32 x 32 (1 x 1) output tiles: same k split and order as the 2 x 2 forms (bit-identical results), more
// blocks for grids that would leave CUs idle.  false: shape not taken (K % 64, alignment).
template <int EPI, int WM, int WN>
bool gemm_launch_small(const GemmArgs &a, hipStream_t st, hipError_t *err)
{
    if (a.K % 64 != 0 || a.ldx % 4 != 0) return false;
    constexpr int KS = 2, BMt = 32 * WM, BNt = 32 * WN;
    const DmaForm f = dma_form<EPI, 1, 1, KS, false, WM, WN>(a.x3 != nullptr);
    dim3 grid((a.N + BNt - 1) / BNt, (a.P + BMt - 1) / BMt);
    GemmArgs args = a;
    const dim3 grid1 = dma_grid((int)grid.x, (int)grid.y, &args);
    void *params[] = {&args};
    *err = hipLaunchKernel(f.fn, grid1, dim3(f.threads), params, f.lds, st);
    return true;
}

template <int EPI>
hipError_t gemm_launch(const GemmArgs &a, hipStream_t st)
{
    // 64 x 64 tiles fill the 256 CUs from N = 4096 at 256 tokens; beyond that 128 x 64 halves
    // the LDS operand reads per MFMA (measured on the 7B shape: 94.7 vs 89.5 TFLOP/s at 512)
    hipError_t e;
    switch (choose_tile(a.N, a.P, false)) {
    case TILE_128x128: return gemm_launch_t<EPI, 2, 2, 2>(a, st);
    case TILE_128x64: return gemm_launch_t<EPI, 2, 1, 2>(a, st);
    case TILE_32x64: if (gemm_launch_small<EPI, 1, 2>(a, st, &e)) return e; break;
    case TILE_32x32: if (gemm_launch_small<EPI, 1, 1>(a, st, &e)) return e; break;
    default: break;
    }
    return gemm_launch_t<EPI, 1, 1, 2>(a, st);
}

// ---- the split-K family (prefill_gemm_dma SPLIT) ----
// Output tile for a product whose K is cut into sk ranges: the same cost model as choose_tile with sk times the
// blocks and 1 / sk of the work per block.  Every form gives the same bits for the same sk.
enum SkTile { SKT_128x64 = 0, SKT_64x64, SKT_32x64 };
static SkTile choose_tile_sk(int N, int P, int sk)
{
    const long long cus = g_cus_hint();
    static const struct { int tok; double eff; } form[3] = {{128, 1.0}, {64, 0.87}, {32, 0.80}};
    int best = -1;
    double best_cost = 0.0;
    for (int f = 0; f < 3; f++) {
        if (f == SKT_128x64 && P <= 64) continue;
        const long long blocks = (long long)((N + 63) / 64) * ((P + form[f].tok - 1) / form[f].tok) * sk;
        const double cost = (double)((blocks + cus - 1) / cus) * (form[f].tok * 64) / sk / form[f].eff;
        if (best < 0 || cost < best_cost * 0.999) {
            best = f;
            best_cost = cost;
        }
    }
    return (SkTile)best;
}

template <int EPI, int TM, int TN, bool PAIR, int WM, int WN>
hipError_t dma_launch_split(GemmArgs a, int n_feat, int sk, const SplitKWs *ws, hipStream_t st)
{
    constexpr int KS = 2, BMt = 32 * WM * TM, BNt = 32 * WN * TN;
    constexpr int feat = PAIR ? BNt / 2 : BNt;  // features (of each matrix when paired) per block
    if (ws == nullptr || ws->part == nullptr || ws->cnt == nullptr) return hipErrorInvalidValue;
    const int ntx = (n_feat + feat - 1) / feat, nty = (a.P + BMt - 1) / BMt;
    if ((size_t)ntx * nty * sk * BMt * BNt > ws->part_floats || ntx * nty > ws->cnt_ints) return hipErrorOutOfMemory;
    a.sk = sk; a.sk_part = ws->part; a.sk_cnt = ws->cnt;
    const DmaForm f = dma_form<EPI, TM, TN, KS, PAIR, WM, WN, 1>(a.x3 != nullptr);
    const dim3 grid = dma_grid(ntx, nty, &a);
    void *params[] = {&a};
    return hipLaunchKernel(f.fn, grid, dim3(f.threads), params, f.lds, st);
}

// unpaired / fused-qkv products (TN = 1 forms) and the paired W1 | W3 product (TN = 2 forms)
template <int EPI, bool PAIR>
hipError_t gemm_launch_sk(const GemmArgs &a, int n_feat, int sk, const SplitKWs *ws, hipStream_t st)
{
    constexpr int TN = PAIR ? 2 : 1;
    switch (choose_tile_sk(n_feat, a.P, sk)) {
    case SKT_128x64: return dma_launch_split<EPI, 2, TN, PAIR, 2, 2>(a, n_feat, sk, ws, st);
    case SKT_64x64: return dma_launch_split<EPI, 1, TN, PAIR, 2, 2>(a, n_feat, sk, ws, st);
    default: return dma_launch_split<EPI, 1, TN, PAIR, 1, 2>(a, n_feat, sk, ws, st);
    }
}

// ---- k-groups on two blocks (prefill_gemm_dma SPLIT == 2): the unsplit family's bits with twice the blocks ----
template <int EPI, int TM, int TN, bool PAIR, int WM, int WN>
hipError_t dma_launch_kgs(GemmArgs a, int n_feat, const SplitKWs *ws, hipStream_t st)
{
    constexpr int KS = 2, BMt = 32 * WM * TM, BNt = 32 * WN * TN;
    constexpr int feat = PAIR ? BNt / 2 : BNt;
    const int ntx = (n_feat + feat - 1) / feat, nty = (a.P + BMt - 1) / BMt;
    if ((size_t)ntx * nty * BMt * BNt > ws->part_floats || 2 * ntx * nty > ws->cnt_ints) return hipErrorOutOfMemory;
    a.sk = 2; a.sk_part = ws->part; a.sk_cnt = ws->cnt;  // sk: the grid's z extent (dma_grid)
    const DmaForm f = dma_form<EPI, TM, TN, KS, PAIR, WM, WN, 2>(a.x3 != nullptr);
    const dim3 grid = dma_grid(ntx, nty, &a);
    void *params[] = {&a};
    return hipLaunchKernel(f.fn, grid, dim3(f.threads), params, f.lds, st);
}

// Whether a [P, N] product (N = features of each matrix when paired) goes out in the two-block form, and on which
// tile.  Same bits either way, so this is grid fill only: the cost model of choose_tile with twice the blocks of
// half the depth, plus the hand-off (~7 us: a dump, a counter, the other block's read) in the model's units.
struct KgsChoice { bool use; TileForm tile; };
static KgsChoice choose_kgs(int N, int P, int K, bool pair, const SplitKWs *ws, bool x3)
{
    KgsChoice none = {false, TILE_64x64};
    if (ws == nullptr || ws->part == nullptr || K % 64 != 0 || x3) return none;   // (the planes form has no two-block variant)
    // Measured (7B shape, whole prefill, interleaved; profiles/r03_prefill_kgs_ab.txt): the form pays where the
    // unsplit family runs ONE 8-wave block of a 128-token tile per CU -- two independent 4-wave blocks of half
    // the LDS hide each other's stage barriers: 512 tokens 58.56 -> 57.32 ms on 128 x 64 tiles -- and loses or
    // ties wherever smaller tiles already put several blocks on a CU (100 ... 300 tokens: -2 ... +4 %; the
    // 110M shape: +10 %).  A cost model of the choose_tile kind picked it for q | k | v and W1 | W3 at 256
    // tokens and lost 5 %: so the rule is the measured one -- chunks of >= 512 tokens, on the 128-token tile the
    // unsplit family takes.
    if (P < 512) return none;
    const TileForm tu = choose_tile(N, P, pair);
    if (tu != TILE_128x64 && tu != TILE_128x128) return none;
    {   // the dump of one k-group per tile and two counters per tile must fit the runstate's workspace (sized for
        // 1024-token chunks: L2Z_PF_CHUNK up to 2048 does not fit) -- else the one-block form, same bits
        const size_t bm = 128, bn = tu == TILE_128x128 ? 128 : (pair ? 128 : 64), feat = pair ? bn / 2 : bn;
        const size_t ntx = ((size_t)N + feat - 1) / feat, nty = ((size_t)P + bm - 1) / bm;
        if (ntx * nty * bm * bn > ws->part_floats || 2 * ntx * nty > (size_t)ws->cnt_ints) return none;
    }
    return {true, tu};
}

// unpaired products with a residual epilogue, the fused q | k | v launch (TN = 1 forms) and the paired W1 | W3 product
template <int EPI, bool PAIR>
hipError_t gemm_launch_kgs(const GemmArgs &a, int n_feat, TileForm tile, const SplitKWs *ws, hipStream_t st)
{
    constexpr int TN = PAIR ? 2 : 1;
    switch (tile) {
    case TILE_128x128:
        if constexpr (!PAIR) return dma_launch_kgs<EPI, 2, 2, false, 2, 2>(a, n_feat, ws, st);
        return hipErrorInvalidValue;
    case TILE_128x64: return dma_launch_kgs<EPI, 2, TN, PAIR, 2, 2>(a, n_feat, ws, st);
    case TILE_64x64: return dma_launch_kgs<EPI, 1, TN, PAIR, 2, 2>(a, n_feat, ws, st);
    default: return dma_launch_kgs<EPI, 1, TN, PAIR, 1, 2>(a, n_feat, ws, st);
    }
}

// The launch's activation matrix as three planes of bf16 terms (prefill_common.h split3): x3[token][plane][kp], element k of
// plane t = term t of x[token][k].  One thread per 8 floats; the columns K .. kp of x are the zero padding (pad_k).
__global__ __launch_bounds__(256) void prefill_split3_kernel(const float *x, int ldx, __bf16 *x3, int kp, int P)
{
    const int per_row = kp >> 3;
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (size_t)P * per_row) return;
    const int p = (int)(id / per_row), c = (int)(id % per_row);
    const v4f *src = (const v4f *)(x + (size_t)p * ldx + 8 * c);
    const Bf3 t = split3(src[0], src[1]);
    v8bf *o = (v8bf *)(x3 + (size_t)p * 3 * kp) + c;
    o[0] = t.t1; o[per_row] = t.t2; o[2 * per_row] = t.t3;
}

// before a launch of the planes form: a.K is the padded K; fills ws->x3 from a.x
hipError_t prepare_x3(GemmArgs &a, const SplitKWs *ws, hipStream_t st, long long n_whole, int planes_ready = PLANES_SPLIT)
{
    if (!x3_applies(n_whole, a.K)) return hipSuccess;   // a.x3 stays null: the f32 matrix cores
    if (ws == nullptr || ws->x3 == nullptr || (size_t)a.P * 3 * a.K * sizeof(__bf16) > ws->x3_bytes) return hipErrorInvalidValue;
    a.x3 = ws->x3; a.kp = a.K; a.ldx3 = 3 * a.K;
    if (planes_ready == PLANES_READY_B) {   // the W1 | W3 launch's epilogue left them in the second planes matrix
        if (ws->x3b == nullptr || (size_t)a.P * 3 * a.K * sizeof(__bf16) > ws->x3b_bytes) return hipErrorInvalidValue;
        a.x3 = ws->x3b;
        return hipSuccess;
    }
    // the launch before this one split the same matrix (k and v behind q), or x's producer wrote them (rmsnorm, attention)
    if (planes_ready == PLANES_READY) return hipSuccess;
    const size_t n = (size_t)a.P * (a.K >> 3);
    prefill_split3_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(a.x, a.ldx, (__bf16 *)ws->x3, a.K, a.P);
    return hipGetLastError();
}

// launch of the stream form; a.K is the padded K, a.N this rank's rows, n_whole the whole model's (the K ranges)
template <int EPI>
hipError_t launch_x3_stream(GemmArgs a, long long n_whole, const SplitKWs *ws, hipStream_t st, int planes_ready = PLANES_SPLIT,
                            DeferredSum *defer = nullptr)
{
    if (defer) defer->valid = false;
    if (const hipError_t e = prepare_x3(a, ws, st, n_whole, planes_ready); e != hipSuccess) return e;
    // Features per block (grid fill only: the K ranges -- the arithmetic -- are x3_stream_sk's whatever the tile).  Chunks of
    // <= 64 tokens may run 192 features on twelve waves or 256 on sixteen instead of 128 on eight: a block streams its W at
    // ~20 GB/s whatever its width (profiles/r06s), so the launch wants as many blocks as fit ONE round -- the narrowest tile
    // that does -- and the wider tiles send the planes through the L2 less often.  (q | k | v: a tile may lie across two of the matrices; a wave's 32 features never do.)  7B shape: q | k | v 192 features (256 blocks of 4 ranges), W1 | W3 192 (230 blocks of 2
    // ranges: 85 us against 105 with 172 blocks of 256 features), wo / W2 128 (256 blocks of 8 ranges).
    const int sk = x3_stream_sk(n_whole, a.P, a.K), tm = (a.P + 31) / 32, cus = g_cus_hint();
    int feat = 128;
    if (x3_stream_tile(a.P) == 256 && tm <= 2 && (long long)((a.N + 127) / 128) * sk > cus) {
        // the NARROWEST tile whose blocks fit one round (the most blocks: the least W per block)
        for (int f : {192, 256}) {
            if (f == 256 && (EPI == G_ROPE || EPI == G_ROPE_CACHE)) continue;   // (sixteen waves of these would spill: 128 registers)
            if ((long long)((a.N + f - 1) / f) * sk <= cus) { feat = f; break; }
        }
    }
    const int ntx = (a.N + feat - 1) / feat;
    if (ws->part == nullptr || ws->cnt == nullptr || (size_t)ntx * sk * tm * 32 * feat > ws->part_floats || 2 * ntx > ws->cnt_ints)
        return hipErrorOutOfMemory;
    a.sk = sk; a.sk_part = ws->part; a.sk_cnt = ws->cnt;
    if (EPI == G_RESID && defer != nullptr && sk > 1 && a.res == a.out && a.ldres == a.ldo) {   // (in place: x += the product)
        a.defer = 1;
        defer->part = ws->part; defer->sk = sk; defer->feat = feat; defer->tm = tm; defer->valid = true;
    }
    // one round of blocks (one per CU: the ring takes the LDS): the ranges of a tile share its reduction and epilogue
    // (same sums in the same order either way: grid fill only, so a rank's own row count decides)
    a.ntx = ntx * sk <= cus ? 0 : ntx;
    const void *fn;
    int nbuf;
    if (feat == 256) {
        if constexpr (EPI == G_ROPE || EPI == G_ROPE_CACHE) return hipErrorInvalidValue;   // (never chosen above)
        else if (tm == 1) { fn = (const void *)prefill_x3_stream<EPI, 1, 4, 8>; nbuf = 4; }
        else { fn = (const void *)prefill_x3_stream<EPI, 2, 3, 8>; nbuf = 3; }
    } else if (feat == 192) {
        if (tm == 1) { fn = (const void *)prefill_x3_stream<EPI, 1, 5, 6>; nbuf = 5; }
        else { fn = (const void *)prefill_x3_stream<EPI, 2, 4, 6>; nbuf = 4; }
    } else {
        switch (tm) {
        case 1: fn = (const void *)prefill_x3_stream<EPI, 1, 7, 4>; nbuf = 7; break;
        case 2: fn = (const void *)prefill_x3_stream<EPI, 2, L2Z_X3_NBUF2, 4>; nbuf = L2Z_X3_NBUF2; break;
        case 3: fn = (const void *)prefill_x3_stream<EPI, 3, 4, 4>; nbuf = 4; break;
        default: fn = (const void *)prefill_x3_stream<EPI, 4, 4, 4>; nbuf = 4; break;
        }
    }
    const size_t lds = (size_t)nbuf * (3 * 32 * tm * 64 + feat * 128);
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    void *params[] = {&a};
    return hipLaunchKernel(fn, dim3((unsigned)(ntx * sk)), dim3((unsigned)(feat * 4)), params, lds, st);
}

}  // namespace

// K ranges per output tile (see l2z_internal.h); > 1 also means "the tile kernel, not the short-prompt kernels".
// Measured on the 7B shape, whole prefill, interleaved (profiles/r03_prefill_splitk_ab.txt):
//   64 tokens   short-prompt kernels 15.8 ms | 64 x 64 tiles unsplit 14.6 | 2 ranges 12.7 | 4 ranges 12.5
//   40 tokens   12.5 | 14.7 | 12.7 | 12.6        100 tokens  unsplit 19.1 | 2 ranges 18.4 | 4 ranges 21.9
//   128 tokens  19.2 | 18.5 | 21.9               200 / 256 tokens: every split slower (29.6 -> 33.8, 31.5 -> 34.0)
// and on the 110M shape (matrices of <= 6 MB, cache resident) every split and the tile kernel below 65 tokens
// lose.  So: matrices that stream from HBM (> 16 MB over the whole model) take 4 ranges at 49 ... 64 tokens --
// where the short-prompt kernels would read W twice -- and the block-starved ones 2 ranges at 65 ... 128;
// everything else stays as it was.  The hand-off (accumulator dump, counter, the last arriver's re-read) costs 6-8 us per launch, which is
// why 4 ranges of a 30-60 us product lose what the larger tile wins.
constexpr long long kRefCus = 256;  // the part the split-K rule was measured on; see prefill_split_k
int prefill_split_k(long long n_whole, int P, int K, bool pair)
{
    if (P > kSplitKMaxTokens) return 1;
    (void)pair;
    int sk = 1;
    {
        const bool streams = n_whole * (long long)K * 4 > ((long long)16 << 20);
        // 65 ... 128 tokens: only the launches the unsplit family leaves block-starved (wo, W2: N = 4096 is 256
        // blocks of 32 x 64, one per CU); q | k | v and W1 | W3 already have ~3 blocks per CU there and LOSE with
        // two ranges (rocprofv3, 128 tokens: wo 61 -> 54 us, W2 162 -> 130, but q|k|v 126 -> 133, W1|W3 237 -> 256)
        const long long unsplit_blocks = ((n_whole + 63) / 64) * ((P + 31) / 32);
        if (streams && P >= 49 && P <= 64) sk = 4;
        // 1.5 blocks per CU of the REFERENCE part (kRefCus = 256: MI355X), as a constant: the split is part of the
        // arithmetic, so it must not depend on the device a rank happens to run on.  (Until round 3 this compared with
        // the running device's CU count: on a part with another CU count the shapes that take two ranges -- and with
        // them the low-order bits of 65 ... 128-token prefills -- differ from results recorded before that change.)
        else if (streams && P > 64 && P <= 128 && 2 * unsplit_blocks <= 3 * kRefCus) sk = 2;
    }
    while (sk > 1 && K % (64 * sk) != 0) sk >>= 1;
    return sk;
}

// out[P,N] = silu(X W1^T) * (X W3^T) in one launch (direct-to-LDS tile kernel, paired form).
// hipErrorNotSupported when the shape does not take that kernel: the caller launches the two GEMMs.
hipError_t launch_prefill_gemm_swiglu_pair(const float *x, int ldx, const float *w1, const float *w3,
                                           float *out, int ldo, int P, int N, int K, hipStream_t st, int n_scale,
                                           int sk, const SplitKWs *ws, int ldw, int planes_ready, int kp_out, bool *planes_written)
{
    if (planes_written) *planes_written = false;
    if (((uintptr_t)x & 15) || ((uintptr_t)w1 & 15) || ((uintptr_t)w3 & 15)) return hipErrorInvalidValue;
    GemmArgs a = {x, w3, w1, out, out, P, N, K, ldx, ldo, ldo, 0, nullptr, 0, n_scale > 0 ? n_scale : 1, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0};
    a.ldw = ldw > 0 ? ldw : K;
    constexpr int skinny_max = Tunables::pf_skinny_max;
    if (const int kp = pad_k(K, 64, ldx); kp > 0 && ldw == 2 * K && w3 == w1 + K &&
        x3_stream_shape(2LL * N * a.n_scale, P, kp)) {
        // the stream form takes W1 | W3 as ONE matrix of alternating rows (the blob's slot)
        GemmArgs b = a;
        b.w = w1; b.w2 = nullptr; b.N = 2 * N; b.K = kp; b.ldw = K;
        // the gated rows' planes beside them (the W2 launch's operand): only whole 64-k rows -- the split launch writes the
        // zeros of pad columns -- and only where the second planes matrix exists (the unsharded pass)
        if (kp_out == N && (N & 63) == 0 && ws != nullptr && ws->x3b != nullptr && (size_t)P * 3 * kp_out * sizeof(__bf16) <= ws->x3b_bytes) {
            b.x3_out = ws->x3b; b.kp_out = kp_out;
            if (planes_written) *planes_written = true;
        }
        return launch_x3_stream<G_SWIGLU_IL>(b, 2LL * N * a.n_scale, ws, st, planes_ready);
    }
    if (P <= skinny_max && sk <= 1) return launch_prefill_skinny_pair(G_SWIGLU, a, st);  // prefill_skinny.hip (or not supported)
    K = a.K = pad_k(K, 64, ldx);   // whole 64-k stages (see pad_k)
    if (K < 0 || ldx % 4 != 0) return hipErrorInvalidValue;
    if (const hipError_t e = prepare_x3(a, ws, st, 2LL * N * a.n_scale, planes_ready); e != hipSuccess) return e;
    // (the gated rows' planes beside them, as in the stream form: every paired tile form finishes in the same epilogue)
    if (a.x3 != nullptr && kp_out == N && (N & 63) == 0 && ws != nullptr && ws->x3b != nullptr && (size_t)P * 3 * kp_out * sizeof(__bf16) <= ws->x3b_bytes) {
        a.x3_out = ws->x3b; a.kp_out = kp_out;
        if (planes_written) *planes_written = true;
    }
    if (sk > 1) return gemm_launch_sk<G_STORE, true>(a, N, sk, ws, st);
    {
        const KgsChoice c = choose_kgs(N, P, K, true, ws, a.x3 != nullptr);
        if (c.use) return gemm_launch_kgs<G_STORE, true>(a, N, c.tile, ws, st);
    }
    constexpr int KS = 2;
    // tokens x (features of W1 + the same features of W3) per block, chosen like the unpaired tiles
    const TileForm tf = choose_tile(N, P, true);
    const int tok = tf == TILE_128x64 ? 128 : tf == TILE_64x64 ? 64 : 32, feat = tf == TILE_32x32 ? 32 : 64;
    const DmaForm f = tf == TILE_128x64 ? dma_form<G_STORE, 2, 2, KS, true>(a.x3 != nullptr)
                    : tf == TILE_64x64  ? dma_form<G_STORE, 1, 2, KS, true>(a.x3 != nullptr)
                    : tf == TILE_32x64  ? dma_form<G_STORE, 1, 2, KS, true, 1, 2>(a.x3 != nullptr)
                                        : dma_form<G_STORE, 1, 2, KS, true, 1, 1>(a.x3 != nullptr);
    const dim3 grid = dma_grid((N + feat - 1) / feat, (P + tok - 1) / tok, &a);
    void *params[] = {&a};
    return hipLaunchKernel(f.fn, grid, dim3(f.threads), params, f.lds, st);
}

// q | k | v of one layer in ONE launch of the direct-to-LDS tile kernel (main.zig:308-358): N = nq + 2 nkv
// features, the block's column range picks the matrix and the epilogue (RoPE into q, RoPE into the key
// cache rows pos0 + token, plain into the value cache rows).  hipErrorNotSupported when the shape does
// not take that kernel or a tile would straddle two ranges: the caller launches the three GEMMs.
hipError_t launch_prefill_gemm_qkv(const float *x, int ldx, const float *wq, const float *wk, const float *wv,
                                   float *q_out, int ldq, float *kcache, float *vcache, int ldkv, int P, int nq,
                                   int nkv, int K, int pos0, const float2 *rope, int head_size, hipStream_t st,
                                   size_t kv_head_stride, int n_scale, int sk, const SplitKWs *ws, int planes_ready)
{
    constexpr int skinny_max = Tunables::pf_skinny_max;
    if (const int kp = pad_k(K, 64, ldx); kp > 0 && x3_stream_shape((long long)(nq + 2 * nkv) * (n_scale > 0 ? n_scale : 1), P, kp)) {
        if (nq % 128 != 0 || nkv % 128 != 0) return hipErrorNotSupported;   // (the caller's three launches take the stream form, same ranges)
        if (((uintptr_t)x & 15) || ((uintptr_t)wq & 15) || ((uintptr_t)wk & 15) || ((uintptr_t)wv & 15)) return hipErrorInvalidValue;
        GemmArgs b = {x, nullptr, wq, q_out, q_out, P, nq + 2 * nkv, kp, ldx, ldq, ldq, pos0, rope, head_size, n_scale > 0 ? n_scale : 1,
                      wk, wv, kcache, vcache, nq, nkv, ldkv, kv_head_stride, 0, 0};
        b.ldw = K;
        return launch_x3_stream<G_QKV>(b, (long long)(nq + 2 * nkv) * b.n_scale, ws, st, planes_ready);
    }
    if (P <= skinny_max && sk <= 1) return hipErrorNotSupported;
    if (((uintptr_t)x & 15) || ((uintptr_t)wq & 15) || ((uintptr_t)wk & 15) || ((uintptr_t)wv & 15)) return hipErrorInvalidValue;
    const int ldw_true = K;        // W rows are K floats apart; the loop runs over whole 64-k stages (see pad_k)
    K = pad_k(K, 64, ldx);
    if (K < 0 || ldx % 4 != 0) return hipErrorInvalidValue;
    const int N = nq + 2 * nkv;
    const long long n_qkv_whole = (long long)N * (n_scale > 0 ? n_scale : 1);
    const bool x3 = x3_applies(n_qkv_whole, K);   // the bf16 matrix cores (prepare_x3 decides the same)
    if (sk > 1) {  // the split family: 64-feature tiles (a tile must not straddle q | k | v)
        if (nq % 64 != 0 || nkv % 64 != 0) return hipErrorNotSupported;
        GemmArgs as = {x, nullptr, wq, q_out, q_out, P, N, K, ldx, ldq, ldq, pos0, rope, head_size, n_scale > 0 ? n_scale : 1,
                       wk, wv, kcache, vcache, nq, nkv, ldkv, kv_head_stride, 0, 0};
        as.ldw = ldw_true;
        if (const hipError_t e = prepare_x3(as, ws, st, (long long)N * as.n_scale, planes_ready); e != hipSuccess) return e;
        return gemm_launch_sk<G_QKV, false>(as, N, sk, ws, st);
    }
    {
        const KgsChoice c = choose_kgs(N, P, K, false, ws, x3);
        const int feat_k = c.tile == TILE_128x128 ? 128 : 64;  // a tile must not straddle q | k | v
        if (c.use && nq % feat_k == 0 && nkv % feat_k == 0) {
            GemmArgs ak = {x, nullptr, wq, q_out, q_out, P, N, K, ldx, ldq, ldq, pos0, rope, head_size, n_scale > 0 ? n_scale : 1,
                           wk, wv, kcache, vcache, nq, nkv, ldkv, kv_head_stride, 0, 0};
            ak.ldw = ldw_true;
            return gemm_launch_kgs<G_QKV, false>(ak, N, c.tile, ws, st);   // (never in the planes mode: choose_kgs)
        }
    }
    TileForm tf = choose_tile(N, P, false);
    // 128 x 64 tiles mean q alone already gives every CU its one resident block: three such launches
    // measured 445 us against 453 for the 768-block one (7B, 512 tokens).  With the smaller tiles several
    // blocks share a CU and the longer grid keeps them supplied: 128 tokens 21.8 -> 19.4 ms fused.
    // (The same on the bf16 cores: 7B, 1024 / 512 tokens 74.3 / 40.6 ms with three launches, 80.3 / 41.6 fused on 128 x 64
    // tiles -- the caller's k and v launches reuse q's planes: planes_ready.)
    (void)x3;
    if (tf == TILE_128x64 || tf == TILE_128x128) return hipErrorNotSupported;
    int feat = tf == TILE_32x32 ? 32 : 64;
    if (nq % feat != 0 || nkv % feat != 0) {
        if (nq % 32 != 0 || nkv % 32 != 0) return hipErrorNotSupported;
        tf = TILE_32x32;
        feat = 32;
    }
    GemmArgs a = {x, nullptr, wq, q_out, q_out, P, N, K, ldx, ldq, ldq, pos0, rope, head_size, 1,
                  wk, wv, kcache, vcache, nq, nkv, ldkv, kv_head_stride, 0, 0};
    a.ldw = ldw_true;
    if (const hipError_t e = prepare_x3(a, ws, st, n_qkv_whole, planes_ready); e != hipSuccess) return e;
    constexpr int KS = 2;
    const int tok = tf == TILE_128x64 ? 128 : tf == TILE_64x64 ? 64 : 32;
    const DmaForm f = tf == TILE_128x64 ? dma_form<G_QKV, 2, 1, KS, false>(a.x3 != nullptr)
                    : tf == TILE_64x64  ? dma_form<G_QKV, 1, 1, KS, false>(a.x3 != nullptr)
                    : tf == TILE_32x64  ? dma_form<G_QKV, 1, 1, KS, false, 1, 2>(a.x3 != nullptr)
                                        : dma_form<G_QKV, 1, 1, KS, false, 1, 1>(a.x3 != nullptr);
    const dim3 grid = dma_grid(N / feat, (P + tok - 1) / tok, &a);
    void *params[] = {&a};
    return hipLaunchKernel(f.fn, grid, dim3(f.threads), params, f.lds, st);
}

// k | v of one layer in ONE launch for short prompts (P <= 64: prefill_skinny.hip's paired form; X is
// brought into the CU once for both).  hipErrorNotSupported otherwise: the caller launches the two.
hipError_t launch_prefill_gemm_kv_pair(const float *x, int ldx, const float *wk, const float *wv, float *kcache,
                                       float *vcache, int ldkv, int P, int nkv, int K, int pos0, const float2 *rope,
                                       int head_size, hipStream_t st, int n_scale, size_t kv_head_stride, int sk,
                                       long long n_launch_whole)
{
    constexpr int skinny_max = Tunables::pf_skinny_max;
    if (P > skinny_max || sk > 1) return hipErrorNotSupported;  // sk > 1: the tile kernel's split-K family takes it
    if (n_launch_whole > 0 && x3_stream_shape(n_launch_whole, P, (K + 63) / 64 * 64)) return hipErrorNotSupported;  // the stream form takes k and v
    if (((uintptr_t)x & 15) || ((uintptr_t)wk & 15) || ((uintptr_t)wv & 15)) return hipErrorInvalidValue;
    GemmArgs a = {x, wv, wk, kcache, kcache, P, nkv, K, ldx, ldkv, ldkv, pos0, rope, head_size, n_scale > 0 ? n_scale : 1,
                  wk, wv, kcache, vcache, 0, nkv, ldkv, kv_head_stride, 0, 0};
    a.ldw = K;
    return launch_prefill_skinny_pair(G_QKV, a, st);
}

// C[P,N] (+)= X[P,K] W[N,K]^T with the chosen epilogue; K % 4 == 0, 16-byte aligned rows
hipError_t launch_prefill_gemm(int epi, const float *x, int ldx, const float *w, float *out, int ldo,
                               int P, int N, int K, int pos0, const float2 *rope, int head_size,
                               hipStream_t st, const float *res, int ldres, int n_scale, size_t kv_head_stride,
                               int sk, const SplitKWs *ws, int ldw, long long n_launch_whole, int planes_ready, DeferredSum *defer)
{
    if (P <= 0 || N <= 0 || K <= 0 || (K % 4) != 0 || (ldx % 4) != 0) return hipErrorInvalidValue;
    if (((uintptr_t)x & 15) || ((uintptr_t)w & 15)) return hipErrorInvalidValue;
    if (defer) defer->valid = false;
    if (res == nullptr) { res = out; ldres = ldo; }  // PG_RESID in place
    GemmArgs a = {x, nullptr, w, out, res, P, N, K, ldx, ldo, ldres, pos0, rope, head_size, n_scale > 0 ? n_scale : 1, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, kv_head_stride, 0, 0};
    a.ldw = ldw > 0 ? ldw : K;
    constexpr int skinny_max = Tunables::pf_skinny_max;
    // n_whole: the rows of the WHOLE model's launch this product belongs to (q | k | v count as one: n_launch_whole)
    if (const int kp = pad_k(K, 64, ldx); kp > 0 && epi != G_SWIGLU &&
        x3_stream_shape(n_launch_whole > 0 ? n_launch_whole : (long long)N * a.n_scale, P, kp)) {
        const long long nw = n_launch_whole > 0 ? n_launch_whole : (long long)N * a.n_scale;
        a.K = kp;
        switch (epi) {
            case G_STORE: return launch_x3_stream<G_STORE>(a, nw, ws, st, planes_ready);
            case G_RESID: return launch_x3_stream<G_RESID>(a, nw, ws, st, planes_ready, defer);
            case G_ROPE: return launch_x3_stream<G_ROPE>(a, nw, ws, st, planes_ready);
            case G_ROPE_CACHE: return launch_x3_stream<G_ROPE_CACHE>(a, nw, ws, st, planes_ready);
            case G_CACHE: return launch_x3_stream<G_CACHE>(a, nw, ws, st, planes_ready);
        }
    }
    if (P <= skinny_max && sk <= 1) return launch_prefill_skinny(epi, a, st);  // prefill_skinny.hip
    K = a.K = pad_k(K, 64, ldx);   // whole 64-k stages (see pad_k)
    if (K < 0) return hipErrorInvalidValue;
    if (const hipError_t e = prepare_x3(a, ws, st, n_launch_whole > 0 ? n_launch_whole : (long long)N * a.n_scale, planes_ready); e != hipSuccess) return e;
    if (sk > 1) {
        if (K / 64 < sk) return hipErrorInvalidValue;
        switch (epi) {
            case G_STORE: return gemm_launch_sk<G_STORE, false>(a, N, sk, ws, st);
            case G_RESID: return gemm_launch_sk<G_RESID, false>(a, N, sk, ws, st);
            case G_ROPE: return gemm_launch_sk<G_ROPE, false>(a, N, sk, ws, st);
            case G_ROPE_CACHE: return gemm_launch_sk<G_ROPE_CACHE, false>(a, N, sk, ws, st);
            case G_CACHE: return gemm_launch_sk<G_CACHE, false>(a, N, sk, ws, st);
            case G_SWIGLU: return gemm_launch_sk<G_SWIGLU, false>(a, N, sk, ws, st);
        }
        return hipErrorInvalidValue;
    }
    if (epi == G_RESID) {  // wo, W2: the two-block form where the grid is short of blocks (same bits)
        const KgsChoice c = choose_kgs(N, P, K, false, ws, a.x3 != nullptr);
        if (c.use) return gemm_launch_kgs<G_RESID, false>(a, N, c.tile, ws, st);
    }
    switch (epi) {
        case G_STORE: return gemm_launch<G_STORE>(a, st);
        case G_RESID: return gemm_launch<G_RESID>(a, st);
        case G_ROPE: return gemm_launch<G_ROPE>(a, st);
        case G_ROPE_CACHE: return gemm_launch<G_ROPE_CACHE>(a, st);
        case G_CACHE: return gemm_launch<G_CACHE>(a, st);
        case G_SWIGLU: return gemm_launch<G_SWIGLU>(a, st);
    }
    return hipErrorInvalidValue;
}

// host logic behind the launches, for tests (include/llama2_hip_test.h): no device needed
int prefill_tile_form(int N, int P, int pair) { return (int)choose_tile(N, P, pair != 0); }

}  // namespace l2z

#ifdef L2Z_X3_TIMELINE
extern "C" int l2z_x3_timeline_dump(long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(l2z::g_xtl), sizeof(long long) * 8 * l2z::kXtlBlocks * 8) != hipSuccess;
}
#endif
