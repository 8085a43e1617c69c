// matvec.hip (with attention.hip, misc_kernels.hip, kernel_common.h) -- hand-written gfx950
// (CDNA4, wave64) kernels for the llama2.zig forward pass.  Every kernel cites the reference lines (src/main.zig) whose
// arithmetic it performs.  Compiled with -ffp-contract=off: every fused
// multiply-add below is an explicit fmaf(), everything else rounds once per
// operation exactly like the reference's scalar code.
//
// Design (DESIGN.md has the numbers):
//  * mat-vec is pure HBM streaming (0.5 flop/byte): one wave owns two weight
//    rows at a time, each lane issues 16-byte non-temporal loads (1 KiB per
//    wave-instruction, 8 in flight per lane), x is staged once per block in
//    LDS, dot products finish with a wave-wide xor-shuffle reduction.  One
//    fixed summation order per output row => results do not depend on grid
//    size or on how rows are sharded over GPUs.
//  * what the reference does immediately before / after each matmul is fused
//    into that launch: rmsnorm (prologue), RoPE + KV-cache write, residual
//    add, SiLU*mul (epilogues).  5 launches per layer.
//  * token and position live in device memory so one captured hipGraph per
//    step can be replayed without host involvement.
// matvec.hip: the fused mat-vec kernels (narrow rows, wide rows, generic scalar) and their launcher.
#include "matvec_device.h"

namespace l2z {
namespace {

// Two dot products against the staged x, generic scalar form (any n / alignment).
__device__ __forceinline__ void dot2_scalar(const float *__restrict__ pa,
                                            const float *__restrict__ pb, const float *xs, int n,
                                            float &ra, float &rb)
{
    const int lane = threadIdx.x & 63;
    float sa = 0.0f, sb = 0.0f;
    for (int j = lane; j < n; j += kWave) {
        const float xv = xs[j];
        sa = fmaf(pa[j], xv, sa);
        sb = fmaf(pb[j], xv, sb);
    }
    ra = wave_sum(sa);
    rb = wave_sum(sb);
}

// ---------------------------------------------------------------------------
// The fused mat-vec.  main.zig:530-605 matmul_fused with its neighbours:
//   PRO_RMS    rmsnorm before it                   :305 / :398 / :426
//   EPI_ROPE   RoPE on q,k + KV-cache row write    :336-358
//   EPI_RESID  accum into the residual stream      :395 / :422
//   EPI_SWIGLU silu(w1.x) * (w3.x)                 :411-416
//
// Work decomposition.  A "pair" is two weight rows that share x reads (rows
// 2p,2p+1 of the concatenated row space -- exactly the RoPE pair (i,i+1) -- or
// row p of w1 and of w3 for SwiGLU).  LPR lanes cooperate on one pair, so a
// wave works on 64/LPR pairs at once:
//   LPR = 64 : large n (7B shapes): one pair per wave, 1 KiB per load instruction
//   LPR < 64 : small n (stories15M/110M): several pairs per wave so that all 64
//              lanes load 16 B and a whole row is in flight at once
// Lane cl of a group takes float4 columns cl, cl+LPR, ... in increasing order
// into 4 component accumulators, then (x+y)+(z+w), then an xor-shuffle over the
// group.  LPR is a function of n only, so a row's summation order never
// depends on the grid, the row count or how rows are sharded over GPUs.
//
// Latency.  Each wave issues the loads of its first weight batch BEFORE the
// block stages x (x's own loads are issued first and return first), so HBM
// latency overlaps the rmsnorm prologue instead of following it.
//
// All kernel arguments are read into scalars and selected with arithmetic:
// indexing the by-value argument block dynamically pushes it into scratch.
// ---------------------------------------------------------------------------
template <int LPR>
struct MvGeom {
    static constexpr int RW = kWave / LPR;          // pairs per wave
    static constexpr int U = (LPR == 64) ? 4 : 6;   // float4 per row per lane per batch
};

// Issue one batch: U float4 of row a and of row b at columns c0 + k*LPR.  Columns past
// the row end are clamped to its last float4: the matching x entries in LDS are the
// zero padding, so they add exactly 0 (weights are finite) -- no predicated loads.
template <int LPR>
__device__ __forceinline__ void mv_load(const float *pa, const float *pb, int c0, int cb, int n4,
                                        v4f (&wa)[MvGeom<LPR>::U], v4f (&wb)[MvGeom<LPR>::U])
{
    constexpr int U = MvGeom<LPR>::U;
    const v4f *a4 = (const v4f *)pa, *b4 = (const v4f *)pb;
    // Per lane: n4 need not be a multiple of the lane count -- the last step of a row like hidden_dim 1376
    // (n4 = 344 = 5 * 64 + 24) is a partial one, the vector form of the reference's scalar tail
    // (main.zig:589-594).
    (void)cb;
#pragma unroll
    for (int k = 0; k < U; k++) {
        int c = c0 + LPR * k;
        c = c < n4 ? c : n4 - 1;
        wa[k] = ldg_nt(a4 + c);
        wb[k] = ldg_nt(b4 + c);
    }
}

template <int LPR>
__device__ __forceinline__ void mv_consume(const v4f *xs4, int c0, const v4f (&wa)[MvGeom<LPR>::U],
                                           const v4f (&wb)[MvGeom<LPR>::U], v4f &acc_a, v4f &acc_b)
{
    constexpr int U = MvGeom<LPR>::U;
#pragma unroll
    for (int k = 0; k < U; k++) {
        const v4f xv = xs4[c0 + LPR * k];  // zero padded to whole batches
        acc_a = fma4(wa[k], xv, acc_a);
        acc_b = fma4(wb[k], xv, acc_b);
    }
}

template <int LPR>
__device__ __forceinline__ float group_sum(float v)
{
    return lanes_sum(v, LPR);
}

template <int PRO, int EPI, int LPR, int XC, bool LL>
__global__ __launch_bounds__(kBlock) void matvec_kernel(const MatvecArgs a)
{
    using G = MvGeom<LPR>;
    constexpr int U = G::U, RW = G::RW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const MvLocals m = mv_locals<EPI>(a);
    const int n4 = m.n >> 2;
    const int n_batches = (n4 + LPR * U - 1) / (LPR * U);
    const int n4_pad = n_batches * (LPR * U);
    float *xs = lds;
    float *scratch = lds + 4 * n4_pad;
    const v4f *xs4 = (const v4f *)xs;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / LPR, cl = lane % LPR;
    const int n_units = (m.n_pairs + RW - 1) / RW;
    const int ustride = gridDim.x * kWaves;

    // 1. issue x loads (they return first), 2. issue the first weight batch, 3. stage x
    v4f xr[LL ? 1 : XC], gr[XC];
    v4u xl[LL ? 2 * XC : 2];
    LLPoll poll;
    if constexpr (LL) {
        poll = ll_poll_init(a.xin);
        xload_issue_ll<PRO, XC>(poll, a.rms_w, n4, xl, gr);
    } else {
        xload_issue<PRO, XC>(a.x, a.rms_w, n4, xr, gr);
    }
    int u = blockIdx.x * kWaves + wave;
    const bool has_unit = u < n_units;
    const float *pa, *pb;
    pair_rows<EPI>(m, (has_unit ? u : 0) * RW + grp, pa, pb);
    v4f wa[U], wb[U];
    EpiIn ein = epi_prefetch<EPI>(m, (has_unit ? u : 0) * RW + grp, cl == 0 && has_unit);
    EpiIn ein_next = ein;
    mv_load<LPR>(pa, pb, cl, 0, n4, wa, wb);
    if constexpr (LL)
        xstage_finish_ll<PRO, XC>(poll, a.rms_w, m.n, n4_pad, xl, gr, xs, scratch);
    else
        xstage_finish<PRO, XC>(a.x, a.rms_w, m.n, n4_pad, xr, gr, xs, scratch);
    if (EPI != EPI_ARGMAX && !has_unit) return;

    // flat loop over (unit, batch): consume the batch in registers, then immediately
    // issue the next one -- the next unit's first batch included -- before reducing
    v4f acc_a = {0.f, 0.f, 0.f, 0.f}, acc_b = {0.f, 0.f, 0.f, 0.f};
    float best_v = -INFINITY;  // EPI_ARGMAX: running (max, first index) of this lane's rows
    int best_i = 0x7fffffff;
    int b = 0;
    while (has_unit) {
        mv_consume<LPR>(xs4, cl + b * (LPR * U), wa, wb, acc_a, acc_b);
        const bool unit_done = (b + 1 == n_batches);
        const int u_next = unit_done ? u + ustride : u;
        const int b_next = unit_done ? 0 : b + 1;
        const bool more = u_next < n_units;
        if (more) {
            if (unit_done) {
                pair_rows<EPI>(m, u_next * RW + grp, pa, pb);
                ein_next = epi_prefetch<EPI>(m, u_next * RW + grp, cl == 0);
            }
            mv_load<LPR>(pa, pb, cl + b_next * (LPR * U), b_next * (LPR * U), n4, wa, wb);
        }
        if (unit_done) {
            const float sa = group_sum<LPR>(hsum4(acc_a));
            const float sb = group_sum<LPR>(hsum4(acc_b));
            pair_epilogue<EPI>(m, u * RW + grp, sa, sb, cl == 0, ein);
            ein = ein_next;
            if (EPI == EPI_ARGMAX) {  // single segment: pair p = rows 2p, 2p+1
                const int ra_ = 2 * (u * RW + grp), rb_ = ra_ + 1;
                if (ra_ < m.total_rows && (sa > best_v || best_i == 0x7fffffff)) {
                    best_v = sa; best_i = ra_ + a.row_offset;
                }
                if (rb_ < m.total_rows && sb > best_v) {  // strict '>' : first index wins ties
                    best_v = sb; best_i = rb_ + a.row_offset;
                }
            }
            acc_a = v4f{0.f, 0.f, 0.f, 0.f};
            acc_b = v4f{0.f, 0.f, 0.f, 0.f};
        }
        if (!more) break;
        u = u_next;
        b = b_next;
    }
    if (EPI == EPI_ARGMAX) {
        // block candidate: larger value wins, equal values -> lower index (main.zig:720)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best_v, o, 64);
            const int oi = __shfl_xor(best_i, o, 64);
            if (oi != 0x7fffffff && (best_i == 0x7fffffff || ov > best_v || (ov == best_v && oi < best_i))) {
                best_v = ov; best_i = oi;
            }
        }
        __syncthreads();
        if (lane == 0) {
            scratch[wave] = best_v;
            scratch[kWaves + wave] = __int_as_float(best_i);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float bv = scratch[0];
            int bi = __float_as_int(scratch[kWaves]);
            for (int w = 1; w < kWaves; w++) {
                const float ov = scratch[w];
                const int oi = __float_as_int(scratch[kWaves + w]);
                if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) {
                    bv = ov; bi = oi;
                }
            }
            a.part_val[blockIdx.x] = bv;
            a.part_idx[blockIdx.x] = bi;
        }
    }
}

// ---------------------------------------------------------------------------
// Wide rows (n >= 4096, n/4 a multiple of 64: the 7B shapes): the WHOLE BLOCK works on
// one pair.  Rows 2p and 2p+1 are adjacent in memory, so a block reads one contiguous
// 8n-byte run per unit and consecutive blocks read consecutive runs -- the chip sweeps
// the matrix linearly, like a plain streaming read (DRAM page locality: +10 % over
// giving every wave its own row pair, measured).  Thread t takes float4 columns
// t, t+256, ...; per-thread component accumulators, (x+y)+(z+w), wave xor-shuffle,
// then the 4 wave partials are added in wave order.  Still a function of n only.
// ---------------------------------------------------------------------------
template <int PRO, int EPI, int XC, bool LL>
__global__ __launch_bounds__(kBlock) void matvec_row_kernel(const MatvecArgs a)
{
    constexpr int U = 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const MvLocals m = mv_locals<EPI>(a);
    const int n4 = m.n >> 2;
    const int n_batches = (n4 + kBlock * U - 1) / (kBlock * U);
    const int n4_pad = n_batches * (kBlock * U);
    float *xs = lds;
    float *scratch = lds + 4 * n4_pad;            // kScratch floats
    float *part = scratch + kScratch;             // [2][kWaves][2] wave partials
    const v4f *xs4 = (const v4f *)xs;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_units = m.n_pairs;
    const int ustride = gridDim.x;

    v4f xr[LL ? 1 : XC], gr[XC];
    v4u xl[LL ? 2 * XC : 2];
    LLPoll poll;
    if constexpr (LL) {
        poll = ll_poll_init(a.xin);
        xload_issue_ll<PRO, XC>(poll, a.rms_w, n4, xl, gr);
    } else {
        xload_issue<PRO, XC>(a.x, a.rms_w, n4, xr, gr);
    }
    int u = blockIdx.x;  // grid <= n_units
    const float *pa, *pb;
    pair_rows<EPI>(m, u, pa, pb);
    v4f wa[U], wb[U];
    auto load = [&](int cb) {  // columns cb + tid + 256k; validity is wave-uniform (n4 % 64 == 0)
        const v4f *a4 = (const v4f *)pa + cb + tid, *b4 = (const v4f *)pb + cb + tid;
        const int wbase = cb + (tid & ~63);
#pragma unroll
        for (int k = 0; k < U; k++) {
            const bool in_row = wbase + kBlock * k < n4;  // wave-uniform
            if (XC == 12 && !in_row) {
                // rows that do not fill their last batch (n = 11008: 704 of 1024 float4): the out-of-row steps
                // load NOTHING (their x is the zero padding) -- the re-reads of the row start were 11.6 % of
                // W2's load instructions and, the nt lines long evicted, 2.7 % extra HBM traffic (PMC 1.027x)
                wa[k] = v4f{0.f, 0.f, 0.f, 0.f};
                wb[k] = v4f{0.f, 0.f, 0.f, 0.f};
                continue;
            }
            const int off = in_row ? kBlock * k : -(cb + (tid & ~63));
            wa[k] = ldg_nt(a4 + off);  // out-of-row steps re-read the row start; their x is 0
            wb[k] = ldg_nt(b4 + off);
        }
    };
    EpiIn ein = epi_prefetch<EPI>(m, u, tid == 0);
    EpiIn ein_next = ein;
    load(0);
    if constexpr (LL)
        xstage_finish_ll<PRO, XC>(poll, a.rms_w, m.n, n4_pad, xl, gr, xs, scratch);
    else
        xstage_finish<PRO, XC>(a.x, a.rms_w, m.n, n4_pad, xr, gr, xs, scratch);

    v4f acc_a = {0.f, 0.f, 0.f, 0.f}, acc_b = {0.f, 0.f, 0.f, 0.f};
    float best_v = -INFINITY;
    int best_i = 0x7fffffff;
    int b = 0, parity = 0;
    while (true) {
#pragma unroll
        for (int k = 0; k < U; k++) {
            const v4f xv = xs4[b * (kBlock * U) + tid + kBlock * k];
            acc_a = fma4(wa[k], xv, acc_a);
            acc_b = fma4(wb[k], xv, acc_b);
        }
        const bool unit_done = (b + 1 == n_batches);
        const int u_next = unit_done ? u + ustride : u;
        const int b_next = unit_done ? 0 : b + 1;
        const bool more = u_next < n_units;
        if (more) {
            if (unit_done) {
                pair_rows<EPI>(m, u_next, pa, pb);
                ein_next = epi_prefetch<EPI>(m, u_next, tid == 0);
            }
            load(b_next * (kBlock * U));
        }
        if (unit_done) {
            const float sa = wave_sum(hsum4(acc_a));
            const float sb = wave_sum(hsum4(acc_b));
            float *pp = part + parity * (2 * kWaves);
            if (lane == 0) {
                pp[wave] = sa;
                pp[kWaves + wave] = sb;
            }
            __syncthreads();
            if (tid == 0) {
                const float ta = ((pp[0] + pp[1]) + pp[2]) + pp[3];
                const float tb = ((pp[kWaves] + pp[kWaves + 1]) + pp[kWaves + 2]) + pp[kWaves + 3];
                pair_epilogue<EPI>(m, u, ta, tb, true, ein);
                if (EPI == EPI_ARGMAX) {
                    const int ra_ = 2 * u, rb_ = ra_ + 1;
                    if (ta > best_v || best_i == 0x7fffffff) { best_v = ta; best_i = ra_ + a.row_offset; }
                    if (rb_ < m.total_rows && tb > best_v) { best_v = tb; best_i = rb_ + a.row_offset; }
                }
            }
            ein = ein_next;
            parity ^= 1;
            acc_a = v4f{0.f, 0.f, 0.f, 0.f};
            acc_b = v4f{0.f, 0.f, 0.f, 0.f};
        }
        if (!more) break;
        u = u_next;
        b = b_next;
    }
    if (EPI == EPI_ARGMAX && tid == 0) {  // units ascend within a block: first index kept
        a.part_val[blockIdx.x] = best_v;
        a.part_idx[blockIdx.x] = best_i;
    }
}

// Generic form: any n, any alignment (the reference's 3x3 / 2x12 known-answer
// tests land here).  One pair per wave, scalar loads.
template <int PRO, int EPI>
__global__ __launch_bounds__(kBlock) void matvec_scalar_kernel(const MatvecArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const MvLocals m = mv_locals<EPI>(a);
    float *xs = lds;
    float *scratch = lds + ((m.n + 3) & ~3);
    stage_x_scalar<PRO>(a.x, a.rms_w, m.n, xs, scratch);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int u = blockIdx.x * kWaves + wave; u < m.n_pairs; u += gridDim.x * kWaves) {
        const float *pa, *pb;
        pair_rows<EPI>(m, u, pa, pb);
        float sa, sb;
        dot2_scalar(pa, pb, xs, m.n, sa, sb);
        pair_epilogue<EPI>(m, u, sa, sb, lane == 0, epi_prefetch<EPI>(m, u, lane == 0));
    }
}

// lanes per pair for a row of n4 float4: the smallest power of two that puts a
// whole row in one batch of 6 loads per lane, clamped to [8, 64].  A function of
// n only (see the kernel header).
int lpr_for(int n4)
{
    int l = 8;
    while (l < 64 && l * 6 < n4) l <<= 1;
    return l;
}

struct MvLaunch {
    const void *fn;
    int lpr, u;
};

template <int PRO, int EPI, int LPR, int XC, bool LL>
MvLaunch mv_entry()
{
    return {reinterpret_cast<const void *>(&matvec_kernel<PRO, EPI, LPR, XC, LL>), LPR, MvGeom<LPR>::U};
}

template <int PRO, int EPI, bool LL>
MvLaunch mv_pick(int lpr, bool big_x)
{
    if (lpr == 8) return mv_entry<PRO, EPI, 8, 4, LL>();
    if (lpr == 16) return mv_entry<PRO, EPI, 16, 4, LL>();
    if (lpr == 32) return mv_entry<PRO, EPI, 32, 4, LL>();
    return big_x ? mv_entry<PRO, EPI, 64, 12, LL>() : mv_entry<PRO, EPI, 64, 4, LL>();
}

template <int PRO, int EPI, bool LL>
const void *mv_row_fn(bool big_x)
{
    return big_x ? reinterpret_cast<const void *>(&matvec_row_kernel<PRO, EPI, 12, LL>)
                 : reinterpret_cast<const void *>(&matvec_row_kernel<PRO, EPI, 4, LL>);
}

// ll: x is read as LL words from the landing slot (sharded runs).  Only the (prologue, epilogue)
// pairs the forward pass launches on a gathered vector exist in that form.
const void *mv_row_pick(int pro, int epi, bool big_x, bool ll)
{
#define L2Z_MVR(P, E) if (pro == P && epi == E && !ll) return mv_row_fn<P, E, false>(big_x);
#define L2Z_MVR_LL(P, E) if (pro == P && epi == E && ll) return mv_row_fn<P, E, true>(big_x);
    L2Z_MVR(PRO_NONE, EPI_STORE)
    L2Z_MVR(PRO_NONE, EPI_RESID)
    L2Z_MVR(PRO_RMS, EPI_STORE)
    L2Z_MVR(PRO_RMS, EPI_ROPE)
    L2Z_MVR(PRO_RMS, EPI_SWIGLU)
    L2Z_MVR(PRO_RMS, EPI_ARGMAX)
    L2Z_MVR_LL(PRO_NONE, EPI_RESID)
    L2Z_MVR_LL(PRO_RMS, EPI_STORE)
    L2Z_MVR_LL(PRO_RMS, EPI_ROPE)
    L2Z_MVR_LL(PRO_RMS, EPI_SWIGLU)
    L2Z_MVR_LL(PRO_RMS, EPI_ARGMAX)   // a shard's classifier on a gathered x: the candidate exchange of greedy steps
#undef L2Z_MVR
#undef L2Z_MVR_LL
    return nullptr;
}

MvLaunch mv_pick_pe(int pro, int epi, int lpr, bool big_x, bool vec, bool ll)
{
#define L2Z_MV(P, E)                                                                      \
    if (pro == P && epi == E && !ll)                                                      \
        return vec ? mv_pick<P, E, false>(lpr, big_x)                                     \
                   : MvLaunch{reinterpret_cast<const void *>(&matvec_scalar_kernel<P, E>), 0, 0};
#define L2Z_MV_LL(P, E) if (pro == P && epi == E && ll && vec) return mv_pick<P, E, true>(lpr, big_x);
    L2Z_MV(PRO_NONE, EPI_STORE)
    L2Z_MV(PRO_NONE, EPI_RESID)
    L2Z_MV(PRO_RMS, EPI_STORE)
    L2Z_MV(PRO_RMS, EPI_ROPE)
    L2Z_MV(PRO_RMS, EPI_SWIGLU)
    L2Z_MV_LL(PRO_NONE, EPI_RESID)
    L2Z_MV_LL(PRO_RMS, EPI_STORE)
    L2Z_MV_LL(PRO_RMS, EPI_ROPE)
    L2Z_MV_LL(PRO_RMS, EPI_SWIGLU)
#undef L2Z_MV
#undef L2Z_MV_LL
    if (pro == PRO_RMS && epi == EPI_ARGMAX && vec && !ll) return mv_pick<PRO_RMS, EPI_ARGMAX, false>(lpr, big_x);
    if (pro == PRO_RMS && epi == EPI_ARGMAX && vec && ll) return mv_pick<PRO_RMS, EPI_ARGMAX, true>(lpr, big_x);
    return {nullptr, 0, 0};
}

// Pure streaming read (non-temporal float4 loads, 8 in flight per lane, sum kept out of DCE's
// reach): the rate the memory system gives a kernel that does nothing else -- the measured
// ceiling the mat-vec GB/s are quoted against beside the 8 TB/s spec (bench.py roofline).
__global__ __launch_bounds__(256) void stream_read_kernel(const v4f *__restrict__ p, size_t n4, float *out)
{
    constexpr int U = 8;
    size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256 * U;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (; i + 256 * (U - 1) < n4; i += stride) {
        v4f r[U];
#pragma unroll
        for (int k = 0; k < U; k++) r[k] = ldg_nt(p + i + 256 * k);
#pragma unroll
        for (int k = 0; k < U; k++) acc += r[k];
    }
    const float s = (acc.x + acc.y) + (acc.z + acc.w);
    if (s == 123.456f) out[blockIdx.x] = s;
}

}  // namespace

// upper bound over every instantiation (n4 padded to whole batches of <= 384 float4)
size_t matvec_lds_bytes(int n) { return (size_t)(4 * ((n >> 2) + 1024) + kScratch + 4 * kWaves + 4) * sizeof(float); }

int matvec_max_grid(int n_cus) { return n_cus * 8; }

// widths the vector kernels take (16-byte aligned operands assumed): every multiple of 4 floats -- a row whose
// float4 count is not a multiple of the lane count ends in a partial step (mv_load).  The rest (n % 4 != 0: the
// reference's 3x3 / 2x12 known-answer tests) goes to the generic scalar kernel, which has no fused-argmax epilogue
bool matvec_vector_width(int n) { return n > 0 && (n % 4) == 0; }

// vector kernels (16-byte aligned operands assumed: every buffer here is a hipMalloc or a row of one)
bool matvec_ll_supported(int n) { return matvec_vector_width(n); }

hipError_t launch_stream_read(const float *p, size_t n_floats, float *out, int n_cus, hipStream_t st)
{
    hipLaunchKernelGGL(stream_read_kernel, dim3(n_cus * 8), dim3(256), 0, st, (const v4f *)p, n_floats / 4, out);
    return hipGetLastError();
}

hipError_t launch_matvec(const MatvecArgs &a_in, int pro, int epi, int max_blocks_per_cu, int n_cus,
                         hipStream_t st, int *out_grid, bool *pushed)
{
    MatvecArgs a = a_in;
    if (max_blocks_per_cu > 8) max_blocks_per_cu = 8;
    bool vec = (a.n % 4) == 0 && aligned16(a.x) && aligned16(a.w0);
    if (a.rows1 > 0) vec = vec && aligned16(a.w1);
    if (a.rows2 > 0) vec = vec && aligned16(a.w2);
    if (pro == PRO_RMS) vec = vec && aligned16(a.rms_w);
    if (epi == EPI_SWIGLU && a.rows1 != a.rows0) return hipErrorInvalidValue;
    const int total_rows = a.rows0 + a.rows1 + a.rows2;
    const int n_pairs = (epi == EPI_SWIGLU) ? a.rows0 : (total_rows + 1) / 2;
    if (n_pairs <= 0 || a.n <= 0) return hipErrorInvalidValue;
    const int n4 = a.n >> 2;
    const int lpr = lpr_for(n4);
    if (epi == EPI_ARGMAX && (!vec || a.rows1 != 0 || a.rows2 != 0)) return hipErrorNotSupported;
    const Tunables &tn = tunables();
    const bool use_row = vec && n4 >= 1024 && (n4 % 64) == 0;
    const bool ll = a.xin.slots != nullptr;
    if (ll && !vec) return hipErrorNotSupported;  // callers ask matvec_ll_supported() first
    MvLaunch k = mv_pick_pe(pro, epi, lpr, a.n > 4096, vec, ll);
    if (use_row) k.fn = mv_row_pick(pro, epi, a.n > 4096, ll);
    if (k.fn == nullptr) return hipErrorInvalidValue;
    size_t lds;
    int n_units;
    if (use_row) {
        const int batch = kBlock * 4;
        const int n4_pad = ((n4 + batch - 1) / batch) * batch;
        lds = (size_t)(4 * n4_pad + kScratch + 4 * kWaves) * sizeof(float);
        n_units = n_pairs;
    } else if (vec) {
        const int batch = k.lpr * k.u;
        const int n4_pad = ((n4 + batch - 1) / batch) * batch;
        lds = (size_t)(4 * n4_pad + kScratch) * sizeof(float);
        const int rw = kWave / k.lpr;
        n_units = (n_pairs + rw - 1) / rw;
    } else {
        lds = (size_t)(((a.n + 3) & ~3) + kScratch) * sizeof(float);
        n_units = n_pairs;
    }
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    // Grid: at most what is resident at once (so every wave's prologue is paid
    // once), units dealt round-robin so every wave gets the same count +-1.
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k.fn, kBlock, lds) != hipSuccess || occ < 1)
        occ = 1;
    if (occ > max_blocks_per_cu) occ = max_blocks_per_cu;
    // The row kernel streams best with few blocks per CU in lock step (fewer concurrent DRAM
    // streams).  Measured at 7B, whole-token rate: 2 blocks/CU 219 tok/s, 1 -> 208, 3 -> 213,
    // 4-8 -> 208-211 (single launches shift against each other under the power cap, so the
    // choice is made on the whole-token rate).
    if (use_row && occ > tn.row_blocks) occ = tn.row_blocks;
    int resident = occ * n_cus;
    // several ranks sharing ONE GPU (tests): a launch must leave room for the peers' kernels it
    // may be waiting for
    if (tn.grid_cap > 0 && resident > tn.grid_cap) resident = tn.grid_cap;
    int grid;
    if (use_row) {  // a unit per block at a time
        grid = n_units;
        if (grid > resident) {
            const int per_block = (n_units + resident - 1) / resident;
            grid = (n_units + per_block - 1) / per_block;
        }
    } else {
        const int blocks_needed = (n_units + kWaves - 1) / kWaves;
        grid = blocks_needed;
        if (grid > resident) {
            const int per_wave = (n_units + resident * kWaves - 1) / (resident * kWaves);
            grid = (n_units + per_wave * kWaves - 1) / (per_wave * kWaves);
        }
    }
    if (out_grid) *out_grid = grid;
    // only single-segment epilogues of the vector kernels push (wo, ffn13, ffn2, classifier)
    if (!vec || epi == EPI_ROPE || a.rows2 != 0 || (epi != EPI_SWIGLU && a.rows1 != 0)) a.push = nullptr;
    if (pushed) *pushed = a.push != nullptr;
    void *args[] = {&a};
    return hipLaunchKernel(k.fn, dim3(grid), dim3(kBlock), args, lds, st);
}

}  // namespace l2z
