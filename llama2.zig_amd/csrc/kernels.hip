// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the llama2.zig
// forward pass.  Every kernel cites the reference lines (src/main.zig) whose
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
#include <cstdlib>

#include "l2z_comm.h"
#include "l2z_internal.h"

namespace l2z {
namespace {

constexpr int kWave = 64;
constexpr int kBlock = 256;
constexpr int kWaves = kBlock / kWave;
constexpr int kScratch = 32;  // floats of LDS scratch for block reductions



typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4f ldg_nt(const v4f *p) { return __builtin_nontemporal_load(p); }

// Cross-lane reductions.  Inside a 16-lane row the exchange is a DPP modifier on a VALU op
// (a few cycles); ds_bpermute-based __shfl_xor (~100 cycles each, and the five steps of one
// sum are a dependent chain) is kept only for the 16- and 32-lane hops.  s_memtime showed the
// shuffle chains were ~2 us of the 8 us attention kernel.  Every lane of the group ends with
// the same value; the order of additions is fixed:
//   xor 1 (quad_perm [1,0,3,2]), xor 2 (quad_perm [2,3,0,1]), 7-i (row_half_mirror),
//   15-i (row_mirror), then xor 16, xor 32.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;

// sum over aligned groups of n lanes (n a power of two, 1..64); all lanes get the result
__device__ __forceinline__ float lanes_sum(float v, int n)
{
    if (n >= 2) v += dpp_mov<kDppXor1>(v);
    if (n >= 4) v += dpp_mov<kDppXor2>(v);
    if (n >= 8) v += dpp_mov<kDppHalfMirror>(v);
    if (n >= 16) v += dpp_mov<kDppMirror>(v);
    if (n >= 32) v += __shfl_xor(v, 16, 64);
    if (n >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

__device__ __forceinline__ float wave_sum(float v) { return lanes_sum(v, 64); }

__device__ __forceinline__ float wave_max(float v)
{
    v = fmaxf(v, dpp_mov<kDppXor1>(v));
    v = fmaxf(v, dpp_mov<kDppXor2>(v));
    v = fmaxf(v, dpp_mov<kDppHalfMirror>(v));
    v = fmaxf(v, dpp_mov<kDppMirror>(v));
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}

// Block-wide reductions: wave shuffle, then the per-wave partials are combined
// by every thread in wave order (fixed order => deterministic).
__device__ __forceinline__ float block_sum(float v, float *scratch)
{
    v = wave_sum(v);
    const int nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = scratch[0];
    for (int i = 1; i < nw; i++) t += scratch[i];
    return t;
}

__device__ __forceinline__ float block_max(float v, float *scratch)
{
    v = wave_max(v);
    const int nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = scratch[0];
    for (int i = 1; i < nw; i++) t = fmaxf(t, scratch[i]);
    return t;
}

__device__ __forceinline__ v4f fma4(v4f a, v4f b, v4f c)
{
    c.x = fmaf(a.x, b.x, c.x);
    c.y = fmaf(a.y, b.y, c.y);
    c.z = fmaf(a.z, b.z, c.z);
    c.w = fmaf(a.w, b.w, c.w);
    return c;
}

__device__ __forceinline__ float hsum4(v4f a) { return (a.x + a.y) + (a.z + a.w); }

// ---------------------------------------------------------------------------
// x staging, optionally with rmsnorm (main.zig:432-468): xs = (x*rsqrt(mean(x^2)+1e-5))*w
// eps is added AFTER the divide by n (:452-453); (x*scale)*w order as :462.
// Generic form (any n, any alignment), used by the scalar kernel and the hooks.
// ---------------------------------------------------------------------------
template <int PRO>
__device__ __forceinline__ void stage_x_scalar(const float *__restrict__ x,
                                               const float *__restrict__ rms_w, int n, float *xs,
                                               float *scratch)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    float ss = 0.0f;
    for (int j = tid; j < n; j += nt) {
        const float v = x[j];
        xs[j] = v;
        if (PRO == PRO_RMS) ss = fmaf(v, v, ss);
    }
    if (PRO == PRO_RMS) {
        const float tot = block_sum(ss, scratch);
        float s = tot / (float)n;
        s += 1e-5f;
        s = 1.0f / sqrtf(s);
        for (int j = tid; j < n; j += nt) xs[j] = (xs[j] * s) * rms_w[j];
    }
    __syncthreads();
}

// Vector form, split in two so the caller can put its first weight loads
// between the halves: xload_issue() only ISSUES the global loads of x (they
// return first: VMEM returns in order), xstage_finish() stores them to LDS,
// normalises and barriers.  XC float4 per thread are held in registers
// (XC*1024 floats); longer x falls back to a load+store loop for the rest.
template <int PRO, int XC>
__device__ __forceinline__ void xload_issue(const float *__restrict__ x,
                                            const float *__restrict__ rms_w, int n4,
                                            v4f (&xr)[XC], v4f (&gr)[XC])
{
    const v4f *x4 = (const v4f *)x;
    const v4f *g4 = (const v4f *)rms_w;
#pragma unroll
    for (int k = 0; k < XC; k++) {
        const int j = threadIdx.x + kBlock * k;
        xr[k] = (j < n4) ? x4[j] : v4f{0.f, 0.f, 0.f, 0.f};
    }
    if (PRO == PRO_RMS) {  // the rmsnorm weights travel with x, ahead of the weight stream
#pragma unroll
        for (int k = 0; k < XC; k++) {
            const int j = threadIdx.x + kBlock * k;
            gr[k] = (j < n4) ? g4[j] : v4f{0.f, 0.f, 0.f, 0.f};
        }
    }
}

template <int PRO, int XC>
__device__ __forceinline__ void xstage_finish(const float *__restrict__ x,
                                              const float *__restrict__ rms_w, int n, int n4_pad,
                                              v4f (&xr)[XC], v4f (&gr)[XC], float *xs,
                                              float *scratch)
{
    const int tid = threadIdx.x;
    const int n4 = n >> 2;
    const v4f *x4 = (const v4f *)x;
    v4f *xs4 = (v4f *)xs;
    float ss = 0.0f;
#pragma unroll
    for (int k = 0; k < XC; k++) {
        const int j = tid + kBlock * k;
        if (j < n4_pad) xs4[j] = xr[k];  // pad region [n4, n4_pad) is zero: 0*w adds nothing
        if (PRO == PRO_RMS) {
            ss = fmaf(xr[k].x, xr[k].x, ss);
            ss = fmaf(xr[k].y, xr[k].y, ss);
            ss = fmaf(xr[k].z, xr[k].z, ss);
            ss = fmaf(xr[k].w, xr[k].w, ss);
        }
    }
    for (int j = tid + kBlock * XC; j < n4_pad; j += kBlock) {  // n > XC*1024 floats
        const v4f v = (j < n4) ? x4[j] : v4f{0.f, 0.f, 0.f, 0.f};
        xs4[j] = v;
        if (PRO == PRO_RMS) {
            ss = fmaf(v.x, v.x, ss);
            ss = fmaf(v.y, v.y, ss);
            ss = fmaf(v.z, v.z, ss);
            ss = fmaf(v.w, v.w, ss);
        }
    }
    if (PRO == PRO_RMS) {
        // scratch is not in use yet: partials -> one barrier -> everyone sums them in wave order
        ss = wave_sum(ss);
        if ((tid & 63) == 0) scratch[tid >> 6] = ss;
        __syncthreads();
        float tot = scratch[0];
#pragma unroll
        for (int i = 1; i < kWaves; i++) tot += scratch[i];
        float s = tot / (float)n;  // :452
        s += 1e-5f;                // :453
        s = 1.0f / sqrtf(s);       // :454
        const v4f *g4 = (const v4f *)rms_w;
#pragma unroll
        for (int k = 0; k < XC; k++) {  // the same j this thread stored above; x, g in registers
            const int j = tid + kBlock * k;
            if (j < n4) {
                v4f v = xr[k];
                v.x = (v.x * s) * gr[k].x;  // :462 values * scale * weights
                v.y = (v.y * s) * gr[k].y;
                v.z = (v.z * s) * gr[k].z;
                v.w = (v.w * s) * gr[k].w;
                xs4[j] = v;
            }
        }
        for (int j = tid + kBlock * XC; j < n4; j += kBlock) {  // n > XC*1024 floats
            v4f v = xs4[j];
            const v4f g = g4[j];
            v.x = (v.x * s) * g.x;
            v.y = (v.y * s) * g.y;
            v.z = (v.z * s) * g.z;
            v.w = (v.w * s) * g.w;
            xs4[j] = v;
        }
    }
    __syncthreads();
}

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
// Kernel arguments copied into plain locals once (keeps them out of scratch).
struct MvLocals {
    const float *w0, *w1, *w2;
    float *out0, *out1, *out2;
    const float *resid;
    const float2 *rope;
    int rows0, r01, total_rows, n_pairs, n, head_size, rope_segs, pos;
    size_t ps1, ps2;
    const P2pArgs *push;  // sharded: LL words of the outputs go straight to the peers
    int push_e;
    size_t push_base;     // index of out0[0] in the gathered vector
};

template <int EPI>
__device__ __forceinline__ MvLocals mv_locals(const MatvecArgs &a)
{
    MvLocals m;
    m.w0 = a.w0; m.w1 = a.w1; m.w2 = a.w2;
    m.out0 = a.out0; m.out1 = a.out1; m.out2 = a.out2;
    m.resid = a.resid; m.rope = a.rope;
    m.rows0 = a.rows0; m.r01 = a.rows0 + a.rows1; m.total_rows = a.rows0 + a.rows1 + a.rows2;
    m.n_pairs = (EPI == EPI_SWIGLU) ? a.rows0 : (m.total_rows + 1) >> 1;
    m.n = a.n; m.head_size = a.head_size; m.rope_segs = a.rope_segs;
    m.pos = (EPI == EPI_ROPE) ? *a.pos_ptr : 0;
    m.ps1 = (size_t)m.pos * (size_t)a.pos_stride1;
    m.ps2 = (size_t)m.pos * (size_t)a.pos_stride2;
    m.push = a.push;
    m.push_e = m.push ? p2p_ll_epoch(m.push) : 0;
    m.push_base = m.push ? (size_t)m.push->rank * m.push->count : 0;
    return m;
}

// the two weight rows of pair p (clamped to the last pair for idle lane groups)
template <int EPI>
__device__ __forceinline__ void pair_rows(const MvLocals &m, int p, const float *&pa,
                                          const float *&pb)
{
    if (p >= m.n_pairs) p = m.n_pairs - 1;
    if (EPI == EPI_SWIGLU) {
        pa = m.w0 + (size_t)p * (size_t)m.n;
        pb = m.w1 + (size_t)p * (size_t)m.n;
    } else {
        const int ga = 2 * p;
        const int gb = (ga + 1 < m.total_rows) ? ga + 1 : ga;
        const bool a1 = ga >= m.rows0, a2 = ga >= m.r01;
        const bool b1 = gb >= m.rows0, b2 = gb >= m.r01;
        const int row_a = ga - (a2 ? m.r01 : (a1 ? m.rows0 : 0));
        const int row_b = gb - (b2 ? m.r01 : (b1 ? m.rows0 : 0));
        const float *wa = a1 ? m.w1 : m.w0;
        wa = a2 ? m.w2 : wa;
        const float *wb = b1 ? m.w1 : m.w0;
        wb = b2 ? m.w2 : wb;
        pa = wa + (size_t)row_a * (size_t)m.n;
        pb = wb + (size_t)row_b * (size_t)m.n;
    }
}

// What the epilogue of pair p reads from memory (residual values, RoPE cos/sin).  Loaded by
// the writer lane when the pair's weight loads are issued, so the epilogue itself never
// waits on memory (a dependent L2 round trip per unit otherwise: ~1 us, serialised).
struct EpiIn {
    float ra, rb;
    float2 cs;
};

template <int EPI>
__device__ __forceinline__ EpiIn epi_prefetch(const MvLocals &m, int p, bool writer)
{
    EpiIn e;
    e.ra = 0.0f; e.rb = 0.0f; e.cs = make_float2(1.0f, 0.0f);
    if (!writer || p >= m.n_pairs) return e;
    if (EPI == EPI_RESID) {  // single segment: rows 2p, 2p+1
        const int ga = 2 * p, gb = ga + 1;
        e.ra = m.resid[ga];
        if (gb < m.total_rows) e.rb = m.resid[gb];
    } else if (EPI == EPI_ROPE) {
        const int ga = 2 * p;
        const bool a1 = ga >= m.rows0, a2 = ga >= m.r01;
        const int seg_a = a2 ? 2 : (a1 ? 1 : 0);
        const int row_a = ga - (a2 ? m.r01 : (a1 ? m.rows0 : 0));
        if (seg_a < m.rope_segs) {
            const int hs = m.head_size;
            e.cs = m.rope[(size_t)m.pos * (size_t)(hs >> 1) + (size_t)((row_a % hs) >> 1)];
        }
    }
    return e;
}

template <int EPI>
__device__ __forceinline__ void pair_epilogue(const MvLocals &m, int p, float sa, float sb,
                                              bool writer, const EpiIn &in)
{
    const bool valid_a = p < m.n_pairs;
    if (EPI == EPI_SWIGLU) {
        float v = sa;
        v = v * (1.0f / (1.0f + expf(-v)));  // :412
        v = v * sb;                          // :416
        if (writer && valid_a) {
            m.out0[p] = v;
            if (m.push) p2p_ll_push(m.push, m.push_e, m.push_base + (size_t)p, v);
        }
        return;
    }
    const int ga = 2 * p, gb = ga + 1;
    const bool valid_b = valid_a && gb < m.total_rows;
    const bool a1 = ga >= m.rows0, a2 = ga >= m.r01;
    const bool b1 = gb >= m.rows0, b2 = gb >= m.r01;
    const int row_a = ga - (a2 ? m.r01 : (a1 ? m.rows0 : 0));
    const int row_b = gb - (b2 ? m.r01 : (b1 ? m.rows0 : 0));
    float *oa = a1 ? m.out1 + m.ps1 : m.out0;
    oa = a2 ? m.out2 + m.ps2 : oa;
    float *ob = b1 ? m.out1 + m.ps1 : m.out0;
    ob = b2 ? m.out2 + m.ps2 : ob;
    if (EPI == EPI_ROPE) {
        // rows (row_a, row_a+1) of one segment: the pair (i, i+1) of :346-349
        float o0 = sa, o1 = sb;
        const int seg_a = a2 ? 2 : (a1 ? 1 : 0);
        if (seg_a < m.rope_segs) {
            const float2 cs = in.cs;     // rope[pos][(row_a % head_size)/2], prefetched
            o0 = sa * cs.x - sb * cs.y;  // :348
            o1 = sa * cs.y + sb * cs.x;  // :349
        }
        if (writer && valid_a) {  // q, or the pos row of the K / V cache (:354-358)
            oa[row_a] = o0;
            if (valid_b) ob[row_b] = o1;
        }
    } else if (EPI == EPI_RESID) {
        if (writer && valid_a) {
            const float va = in.ra + sa, vb = in.rb + sb;  // :711 a[i] += b[i]  (resid[row] prefetched)
            oa[row_a] = va;
            if (valid_b) ob[row_b] = vb;
            if (m.push) {  // single segment on this path: row == index in the slice
                p2p_ll_push(m.push, m.push_e, m.push_base + (size_t)row_a, va);
                if (valid_b) p2p_ll_push(m.push, m.push_e, m.push_base + (size_t)row_b, vb);
            }
        }
    } else {
        if (writer && valid_a) {
            oa[row_a] = sa;
            if (valid_b) ob[row_b] = sb;
            if (m.push) {
                p2p_ll_push(m.push, m.push_e, m.push_base + (size_t)row_a, sa);
                if (valid_b) p2p_ll_push(m.push, m.push_e, m.push_base + (size_t)row_b, sb);
            }
        }
    }
}

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
    if (LPR == 64) {
        // n4 % 64 == 0 (checked by the launcher): whether step k of the batch that starts
        // at column cb is inside the row is the same for every lane.  Out-of-row steps
        // re-read step 0 (their x entries are the zero padding).
#pragma unroll
        for (int k = 0; k < U; k++) {
            const int off = (cb + 64 * k < n4) ? 64 * k : 0;  // wave-uniform
            wa[k] = ldg_nt(a4 + c0 + off);
            wb[k] = ldg_nt(b4 + c0 + off);
        }
    } else {
#pragma unroll
        for (int k = 0; k < U; k++) {
            int c = c0 + LPR * k;
            c = c < n4 ? c : n4 - 1;
            wa[k] = ldg_nt(a4 + c);
            wb[k] = ldg_nt(b4 + c);
        }
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

template <int PRO, int EPI, int LPR, int XC>
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
    v4f xr[XC], gr[XC];
    xload_issue<PRO, XC>(a.x, a.rms_w, n4, xr, gr);
    int u = blockIdx.x * kWaves + wave;
    const bool has_unit = u < n_units;
    const float *pa, *pb;
    pair_rows<EPI>(m, (has_unit ? u : 0) * RW + grp, pa, pb);
    v4f wa[U], wb[U];
    EpiIn ein = epi_prefetch<EPI>(m, (has_unit ? u : 0) * RW + grp, cl == 0 && has_unit);
    EpiIn ein_next = ein;
    mv_load<LPR>(pa, pb, cl, 0, n4, wa, wb);
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
template <int PRO, int EPI, int XC>
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

    v4f xr[XC], gr[XC];
    xload_issue<PRO, XC>(a.x, a.rms_w, n4, xr, gr);
    int u = blockIdx.x;  // grid <= n_units
    const float *pa, *pb;
    pair_rows<EPI>(m, u, pa, pb);
    v4f wa[U], wb[U];
    auto load = [&](int cb) {  // columns cb + tid + 256k; validity is wave-uniform (n4 % 64 == 0)
        const v4f *a4 = (const v4f *)pa + cb + tid, *b4 = (const v4f *)pb + cb + tid;
        const int wbase = cb + (tid & ~63);
#pragma unroll
        for (int k = 0; k < U; k++) {
            const int off = (wbase + kBlock * k < n4) ? kBlock * k : -(cb + (tid & ~63));
            wa[k] = ldg_nt(a4 + off);  // out-of-row steps re-read the row start; their x is 0
            wb[k] = ldg_nt(b4 + off);
        }
    };
    EpiIn ein = epi_prefetch<EPI>(m, u, tid == 0);
    EpiIn ein_next = ein;
    load(0);
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
#ifdef L2Z_DBG_NOREDUCE
            const float sa = hsum4(acc_a), sb = hsum4(acc_b);
#else
            const float sa = wave_sum(hsum4(acc_a));
            const float sb = wave_sum(hsum4(acc_b));
#endif
            float *pp = part + parity * (2 * kWaves);
            if (lane == 0) {
                pp[wave] = sa;
                pp[kWaves + wave] = sb;
            }
#ifndef L2Z_DBG_NOBARRIER
            __syncthreads();
#endif
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

// ---------------------------------------------------------------------------
// Attention for one head per block (main.zig:361-389).
// Thread (g, c): group g of TPR lanes walks timesteps t = g, g+G, ...; lane c
// owns float4 column(s) c of the head.
// ---------------------------------------------------------------------------
struct AttnGeom {
    int E;    // elements per head row in load units (head_size/4 if VEC else head_size)
    int TPR;  // lanes per row: power of two, <= 64
    int G;    // groups per block
};

__host__ __device__ inline AttnGeom attn_geom(int head_size, bool vec, int block = kBlock)
{
    AttnGeom g;
    g.E = vec ? head_size >> 2 : head_size;
    int t = 1;
    while (t < g.E && t < 64) t <<= 1;
    g.TPR = t;
    g.G = block / t;
    return g;
}

// scores for timesteps t < T: att[t] = dot(q, K[t]) / div   (:367-375).
// Group g walks t = g, g+G, ...; kAttnUB timesteps are loaded before any is
// used so kAttnUB K rows are in flight per lane (the first build did one
// dependent load per step: 16 serial round trips at pos 255).
constexpr int kAttnUB = 4;

template <bool VEC>
__device__ __forceinline__ void attn_scores(const float *qs, const float *__restrict__ kbase,
                                            int kv_stride, int head_size, int T, float div,
                                            float *att)
{
    const AttnGeom ge = attn_geom(head_size, VEC);
    const int g = threadIdx.x / ge.TPR, c0 = threadIdx.x % ge.TPR;
    for (int t0 = g; t0 < T; t0 += ge.G * kAttnUB) {
        float p[kAttnUB];
#pragma unroll
        for (int i = 0; i < kAttnUB; i++) p[i] = 0.0f;
        for (int c = c0; c < ge.E; c += ge.TPR) {  // one trip unless head_size > 256
            if (VEC) {
                v4f kv[kAttnUB];
#pragma unroll
                for (int i = 0; i < kAttnUB; i++) {
                    int t = t0 + ge.G * i;
                    t = t < T ? t : T - 1;  // clamped: result discarded below
                    kv[i] = ((const v4f *)(kbase + (size_t)t * (size_t)kv_stride))[c];
                }
                const v4f qv = ((const v4f *)qs)[c];
#pragma unroll
                for (int i = 0; i < kAttnUB; i++) {
                    v4f acc = {0.f, 0.f, 0.f, 0.f};
                    acc = fma4(qv, kv[i], acc);
                    p[i] += hsum4(acc);
                }
            } else {
                float kv[kAttnUB];
#pragma unroll
                for (int i = 0; i < kAttnUB; i++) {
                    int t = t0 + ge.G * i;
                    t = t < T ? t : T - 1;
                    kv[i] = kbase[(size_t)t * (size_t)kv_stride + c];
                }
                const float qv = qs[c];
#pragma unroll
                for (int i = 0; i < kAttnUB; i++) p[i] = fmaf(qv, kv[i], p[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < kAttnUB; i++) {
            float v = p[i];
            v = lanes_sum(v, ge.TPR);
            const int t = t0 + ge.G * i;
            if (c0 == 0 && t < T) att[t] = v / div;  // :372 divide, not multiply by reciprocal
        }
    }
}

// in-place softmax over att[0..T)  (main.zig:687-706)
__device__ __forceinline__ void block_softmax(float *att, int T, float *scratch)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    float m = -INFINITY;
    for (int t = tid; t < T; t += nt) m = fmaxf(m, att[t]);
    m = block_max(m, scratch);
    float s = 0.0f;
    for (int t = tid; t < T; t += nt) {
        const float e = expf(att[t] - m);  // :699
        att[t] = e;
        s += e;
    }
    s = block_sum(s, scratch);
    for (int t = tid; t < T; t += nt) att[t] = att[t] / s;  // :704 divide
    __syncthreads();
}

// out[i] = sum_t att[t] * V[t][i]   (main.zig:657-685): G interleaved partial
// sums per column (t = g, g+G, ... in increasing t), combined in g order.
// kAttnUB V rows are loaded ahead of their use.
template <bool VEC>
__device__ __forceinline__ void attn_weighted_sum(const float *att, const float *__restrict__ vbase,
                                                  int kv_stride, int head_size, int T, float *part,
                                                  float *out)
{
    const AttnGeom ge = attn_geom(head_size, VEC);
    const int g = threadIdx.x / ge.TPR, c0 = threadIdx.x % ge.TPR;
    for (int c = c0; c < ge.E; c += ge.TPR) {
        if (VEC) {
            v4f acc = {0.f, 0.f, 0.f, 0.f};
            for (int t0 = g; t0 < T; t0 += ge.G * kAttnUB) {
                v4f vv[kAttnUB];
                float w[kAttnUB];
#pragma unroll
                for (int i = 0; i < kAttnUB; i++) {
                    const int t = t0 + ge.G * i;
                    const int tc = t < T ? t : T - 1;
                    vv[i] = ((const v4f *)(vbase + (size_t)tc * (size_t)kv_stride))[c];
                    w[i] = t < T ? att[tc] : 0.0f;
                }
#pragma unroll
                for (int i = 0; i < kAttnUB; i++) {  // increasing t
                    acc.x = fmaf(vv[i].x, w[i], acc.x);
                    acc.y = fmaf(vv[i].y, w[i], acc.y);
                    acc.z = fmaf(vv[i].z, w[i], acc.z);
                    acc.w = fmaf(vv[i].w, w[i], acc.w);
                }
            }
            ((v4f *)(part + (size_t)g * head_size))[c] = acc;
        } else {
            float acc = 0.0f;
            for (int t0 = g; t0 < T; t0 += ge.G * kAttnUB) {
                float vv[kAttnUB], w[kAttnUB];
#pragma unroll
                for (int i = 0; i < kAttnUB; i++) {
                    const int t = t0 + ge.G * i;
                    const int tc = t < T ? t : T - 1;
                    vv[i] = vbase[(size_t)tc * (size_t)kv_stride + c];
                    w[i] = t < T ? att[tc] : 0.0f;
                }
#pragma unroll
                for (int i = 0; i < kAttnUB; i++) acc = fmaf(vv[i], w[i], acc);
            }
            part[(size_t)g * head_size + c] = acc;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < head_size; i += blockDim.x) {
        float s = part[i];
        for (int gg = 1; gg < ge.G; gg++) s += part[(size_t)gg * head_size + i];
        out[i] = s;
    }
}

// Softmax over att[0..T) (main.zig:687-706) computed redundantly by every wave --
// each wave reduces max and sum over ALL T with the same instruction sequence, so
// all waves hold bit-identical (max, sum) without any cross-wave barrier -- and
// wave w normalises the entries t = w*64 + lane, + blockDim, ...
// The normalised weights go to a second buffer, so no wave overwrites what another still reads.
__device__ __forceinline__ void wave_softmax(const float *att, float *prob, int T)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    float m = -INFINITY;
    for (int t = lane; t < T; t += kWave) m = fmaxf(m, att[t]);
    m = wave_max(m);
    float s = 0.0f;
    for (int t = lane; t < T; t += kWave) s += expf(att[t] - m);  // :699
    s = wave_sum(s);
    for (int t = wave * kWave + lane; t < T; t += nw * kWave) prob[t] = expf(att[t] - m) / s;  // :704
    __syncthreads();
}

// out[i] = part[0][i] + part[1][i] + ... + part[G-1][i], i < hs.  R = 1..16 adjacent lanes
// share one output: lane r adds partials r, r+R, ... (increasing), then a DPP sum over the R
// lanes.  R depends only on (G, hs, blockDim) -- fixed per model.
__device__ __forceinline__ void reduce_partials(const float *part, int G, int hs, float *out,
                                                const P2pArgs *push = nullptr, size_t push_idx0 = 0)
{
    int R = 1;
    while (R * 2 <= G && R * 2 * hs <= (int)blockDim.x && R < 16) R <<= 1;
    const int i = threadIdx.x / R, r = threadIdx.x % R;
    float s = 0.0f;
    if (i < hs)
        for (int gg = r; gg < G; gg += R) s += part[(size_t)gg * hs + i];
    s = lanes_sum(s, R);
    if (i < hs && r == 0) {
        out[i] = s;
        if (push) p2p_ll_push(push, p2p_ll_epoch(push), push_idx0 + (size_t)i, s);
    }
}

// Fast path (head_size % 4 == 0, head_size <= 256).  The first kFastUB timesteps of
// every group -- K rows AND V rows -- are requested up front, WITHOUT waiting for pos:
// rows past pos exist (the cache has seq_len rows, zero-initialised or holding finite
// values of an earlier sequence) and are masked, so pos, q, K and V travel in one
// round trip and the V rows arrive while the softmax runs.  Same arithmetic and
// summation order as attn_scores / attn_weighted_sum.  Kept compact on purpose: at
// stories15M sizes this kernel's time is launch + instruction fetch, not data.
constexpr int kFastUB = 8;
constexpr int kAttnFastBlock = 1024;  // long contexts: 16 waves per head (32 groups at head_size 128)

// NT = 256 for short contexts (seq_len <= 512: launch latency matters most),
// NT = 1024 for long ones (more rows in flight per head).
#ifdef L2Z_DBG_TS
__device__ long long g_dbg_ts[16];
#define L2Z_TS(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_dbg_ts[i] = clock64(); } while (0)
#else
#define L2Z_TS(i) do { } while (0)
#endif
template <int NT, bool SPEC>
__global__ __launch_bounds__(NT) void attention_fast_kernel(const AttnArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int hs = a.head_size;
    const AttnGeom ge = attn_geom(hs, true, NT);
    float *att = lds;                                  // seq_len raw scores
    float *prob = att + ((a.seq_len + 3) & ~3);        // seq_len softmax weights
    float *part = prob + ((a.seq_len + 3) & ~3);       // G*hs
    const int h = blockIdx.x;
    const int kvh = h / a.kv_mul;                      // :369 (h / kv_mul) * head_size
    const float *kbase = a.kcache + (size_t)kvh * hs;
    const float *vbase = a.vcache + (size_t)kvh * hs;
    const size_t stride = (size_t)a.kv_dim;
    const int g = threadIdx.x / ge.TPR, c0 = threadIdx.x % ge.TPR;
    const bool active = c0 < ge.E;
    const int cc = active ? c0 : 0;
    const int step = ge.G * kFastUB;
    const v4f zero = {0.f, 0.f, 0.f, 0.f};

    L2Z_TS(0);
    // SPEC (small models, latency-bound): the first round is requested without waiting for
    // pos -- rows past pos exist and are masked -- so pos, q, K and V travel in one round trip.
    // !SPEC (large heads): one CU pulls only ~45 GB/s, speculative rows would cost more than the
    // extra dependent read of pos, so rows are clamped to pos (duplicates hit the L1).
    const int T = *a.pos_ptr + 1;  // timesteps 0..pos inclusive (:367)
    const int lim = SPEC ? a.seq_len : T;
    const v4f qv = active ? ((const v4f *)(a.q + (size_t)h * hs))[cc] : zero;
    v4f kr[kFastUB], vr[kFastUB];
#pragma unroll
    for (int i = 0; i < kFastUB; i++) {
        int t = g + ge.G * i;
        t = t < lim ? t : lim - 1;
        kr[i] = ((const v4f *)(kbase + (size_t)t * stride))[cc];
    }
#pragma unroll
    for (int i = 0; i < kFastUB; i++) {
        int t = g + ge.G * i;
        t = t < lim ? t : lim - 1;
        vr[i] = ((const v4f *)(vbase + (size_t)t * stride))[cc];
    }
    L2Z_TS(1);
    const float div = sqrtf((float)hs);
    for (int t0 = g;;) {  // scores (:367-375)
#pragma unroll
        for (int i = 0; i < kFastUB; i++) {
            float p = hsum4(fma4(qv, kr[i], zero));
            p = lanes_sum(p, ge.TPR);
            const int t = t0 + ge.G * i;
            if (c0 == 0 && t < T) att[t] = p / div;  // :372 divide
        }
        t0 += step;
        if (t0 >= T) break;
#pragma unroll
        for (int i = 0; i < kFastUB; i++) {
            int t = t0 + ge.G * i;
            t = t < T ? t : T - 1;
            kr[i] = ((const v4f *)(kbase + (size_t)t * stride))[cc];
        }
    }
    L2Z_TS(2);
    __syncthreads();
    L2Z_TS(3);
    wave_softmax(att, prob, T);  // :378
    L2Z_TS(4);
    v4f acc = zero;
    for (int t0 = g;;) {  // att . V (:381-388), increasing t within the group
#pragma unroll
        for (int i = 0; i < kFastUB; i++) {
            const int t = t0 + ge.G * i;
            const float w = t < T ? prob[t] : 0.0f;
            acc.x = fmaf(vr[i].x, w, acc.x);
            acc.y = fmaf(vr[i].y, w, acc.y);
            acc.z = fmaf(vr[i].z, w, acc.z);
            acc.w = fmaf(vr[i].w, w, acc.w);
        }
        t0 += step;
        if (t0 >= T) break;
#pragma unroll
        for (int i = 0; i < kFastUB; i++) {
            int t = t0 + ge.G * i;
            t = t < T ? t : T - 1;
            vr[i] = ((const v4f *)(vbase + (size_t)t * stride))[cc];
        }
    }
    L2Z_TS(5);
    if (active) ((v4f *)(part + (size_t)g * hs))[cc] = acc;
    __syncthreads();
    L2Z_TS(6);
    // sharded: a.xb already points at this rank's slice, head h of it starts at h * hs
    reduce_partials(part, ge.G, hs, a.xb + (size_t)h * hs, a.push,
                    a.push ? (size_t)a.push->rank * a.push->count + (size_t)h * hs : 0);
    L2Z_TS(7);
}

// ---------------------------------------------------------------------------
// Split attention (flash-decoding form).  One CU pulls only ~45 GB/s, so one block per
// head (attention_fast_kernel) leaves 7/8 of a 256-CU chip idle at 32 heads and spends its
// time waiting for its own K/V rows (measured with s_memtime: 6 of 10 us).  Here head h is
// shared by `nch` blocks; block (h, c) owns timesteps t = c, c+nch, c+2*nch, ... and writes
//     m_c = max score,  l_c = sum exp(score - m_c),  o_c[i] = sum exp(score - m_c) * V[t][i]
// attention_combine_kernel then forms  out[i] = (sum_c o_c[i] e^(m_c-M)) / (sum_c l_c e^(m_c-M)),
// M = max_c m_c.  Mathematically main.zig:361-389; in floating point the weights are
// e^(s-m_c) * e^(m_c-M) / L instead of e^(s-M) / L (a few ulp), well inside the logit
// tolerance, and independent of GPU count (attention is head-local).
// part layout: [head][chunk][head_size + 4] floats = o_c[head_size], m_c, l_c, pad, pad
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void attention_split_kernel(const AttnArgs a, int nch,
                                                                 float *__restrict__ part_out)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int hs = a.head_size;
    const AttnGeom ge = attn_geom(hs, true, kBlock);
    const int max_local = (a.seq_len + nch - 1) / nch;
    float *sc = lds;                                   // local scores
    float *wt = sc + ((max_local + 3) & ~3);           // local unnormalised weights
    float *part = wt + ((max_local + 3) & ~3);         // G*hs
    const int h = blockIdx.x / nch, c = blockIdx.x % nch;
    const int kvh = h / a.kv_mul;                      // :369
    const float *kbase = a.kcache + (size_t)kvh * hs;
    const float *vbase = a.vcache + (size_t)kvh * hs;
    const size_t stride = (size_t)a.kv_dim;
    const int g = threadIdx.x / ge.TPR, c0 = threadIdx.x % ge.TPR;
    const bool active = c0 < ge.E;
    const int cc = active ? c0 : 0;
    const int step = ge.G * kFastUB;
    const v4f zero = {0.f, 0.f, 0.f, 0.f};

    const int T = *a.pos_ptr + 1;                      // :367
    const int Tc = T > c ? (T - c + nch - 1) / nch : 0;  // timesteps owned by this block
    float *po = part_out + ((size_t)h * nch + c) * (size_t)(hs + 4);
    if (Tc == 0) {  // pos < c: empty chunk (uniform branch)
        for (int i = threadIdx.x; i < hs; i += blockDim.x) po[i] = 0.0f;
        if (threadIdx.x == 0) { po[hs] = -INFINITY; po[hs + 1] = 0.0f; }
        return;
    }
    const v4f qv = active ? ((const v4f *)(a.q + (size_t)h * hs))[cc] : zero;
    v4f kr[kFastUB], vr[kFastUB];
#pragma unroll
    for (int i = 0; i < kFastUB; i++) {  // rows clamped to the chunk's last one (duplicates hit L1)
        int j = g + ge.G * i;
        j = j < Tc ? j : Tc - 1;
        kr[i] = ((const v4f *)(kbase + (size_t)(c + nch * j) * stride))[cc];
    }
#pragma unroll
    for (int i = 0; i < kFastUB; i++) {
        int j = g + ge.G * i;
        j = j < Tc ? j : Tc - 1;
        vr[i] = ((const v4f *)(vbase + (size_t)(c + nch * j) * stride))[cc];
    }
    const float div = sqrtf((float)hs);
    for (int j0 = g;;) {  // scores (:367-375), local index j <-> t = c + nch*j
#pragma unroll
        for (int i = 0; i < kFastUB; i++) {
            float p = hsum4(fma4(qv, kr[i], zero));
            p = lanes_sum(p, ge.TPR);
            const int j = j0 + ge.G * i;
            if (c0 == 0 && j < Tc) sc[j] = p / div;  // :372
        }
        j0 += step;
        if (j0 >= Tc) break;
#pragma unroll
        for (int i = 0; i < kFastUB; i++) {
            int j = j0 + ge.G * i;
            j = j < Tc ? j : Tc - 1;
            kr[i] = ((const v4f *)(kbase + (size_t)(c + nch * j) * stride))[cc];
        }
    }
    __syncthreads();
    // chunk-local max and sum of exponentials, redundantly per wave (identical in every wave)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float m = -INFINITY;
    for (int j = lane; j < Tc; j += kWave) m = fmaxf(m, sc[j]);
    m = wave_max(m);
    float l = 0.0f;
    for (int j = lane; j < Tc; j += kWave) l += expf(sc[j] - m);
    l = wave_sum(l);
    for (int j = wave * kWave + lane; j < Tc; j += kBlock) wt[j] = expf(sc[j] - m);  // unnormalised
    __syncthreads();
    v4f acc = zero;
    for (int j0 = g;;) {  // weighted V (:381-388), increasing t within the group
#pragma unroll
        for (int i = 0; i < kFastUB; i++) {
            const int j = j0 + ge.G * i;
            const float w = j < Tc ? wt[j] : 0.0f;
            acc.x = fmaf(vr[i].x, w, acc.x);
            acc.y = fmaf(vr[i].y, w, acc.y);
            acc.z = fmaf(vr[i].z, w, acc.z);
            acc.w = fmaf(vr[i].w, w, acc.w);
        }
        j0 += step;
        if (j0 >= Tc) break;
#pragma unroll
        for (int i = 0; i < kFastUB; i++) {
            int j = j0 + ge.G * i;
            j = j < Tc ? j : Tc - 1;
            vr[i] = ((const v4f *)(vbase + (size_t)(c + nch * j) * stride))[cc];
        }
    }
    if (active) ((v4f *)(part + (size_t)g * hs))[cc] = acc;
    __syncthreads();
    reduce_partials(part, ge.G, hs, po);
    if (threadIdx.x == 0) { po[hs] = m; po[hs + 1] = l; }
}

__global__ void attention_combine_kernel(const float *__restrict__ part_in, int nch, int hs,
                                         float *__restrict__ xb, const P2pArgs *push)
{
    constexpr int kMaxCh = 16;
    const int h = blockIdx.x;
    const float *p = part_in + (size_t)h * nch * (size_t)(hs + 4);
    float mc[kMaxCh], lc[kMaxCh];
#pragma unroll
    for (int c = 0; c < kMaxCh; c++) {  // all chunk statistics in one round trip
        const int cc = c < nch ? c : 0;
        mc[c] = p[(size_t)cc * (hs + 4) + hs];
        lc[c] = p[(size_t)cc * (hs + 4) + hs + 1];
    }
    float M = -INFINITY;
#pragma unroll
    for (int c = 0; c < kMaxCh; c++)
        if (c < nch) M = fmaxf(M, mc[c]);
    float den = 0.0f;
#pragma unroll
    for (int c = 0; c < kMaxCh; c++) {
        mc[c] = c < nch ? expf(mc[c] - M) : 0.0f;  // scale of chunk c; empty chunk: e^(-inf) = 0
        den = fmaf(lc[c], mc[c], den);
    }
    for (int i = threadIdx.x; i < hs; i += blockDim.x) {
        float oc[kMaxCh];
#pragma unroll
        for (int c = 0; c < kMaxCh; c++) oc[c] = p[(size_t)(c < nch ? c : 0) * (hs + 4) + i];
        float num = 0.0f;
#pragma unroll
        for (int c = 0; c < kMaxCh; c++) num = fmaf(oc[c], mc[c], num);
        const float v = num / den;
        xb[(size_t)h * hs + i] = v;
        if (push)
            p2p_ll_push(push, p2p_ll_epoch(push), (size_t)push->rank * push->count + (size_t)h * hs + i, v);
    }
}

// Generic path: any head_size / alignment.
template <bool VEC>
__global__ __launch_bounds__(kBlock) void attention_kernel(const AttnArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int hs = a.head_size;
    const AttnGeom ge = attn_geom(hs, VEC);
    float *qs = lds;                                  // hs
    float *att = qs + ((hs + 3) & ~3);                // seq_len
    float *part = att + ((a.seq_len + 3) & ~3);       // G*hs
    float *scratch = part + (size_t)ge.G * hs;        // kScratch

    const int h = blockIdx.x;
    const int kvh = h / a.kv_mul;                     // :369 (h / kv_mul) * head_size
    const int T = *a.pos_ptr + 1;                     // timesteps 0..pos inclusive (:367)
    for (int i = threadIdx.x; i < hs; i += blockDim.x) qs[i] = a.q[(size_t)h * hs + i];
    __syncthreads();
    attn_scores<VEC>(qs, a.kcache + (size_t)kvh * hs, a.kv_dim, hs, T, sqrtf((float)hs), att);
    __syncthreads();
    block_softmax(att, T, scratch);                   // :378
    attn_weighted_sum<VEC>(att, a.vcache + (size_t)kvh * hs, a.kv_dim, hs, T, part,
                           a.xb + (size_t)h * hs);    // :381-388
}

// ---------------------------------------------------------------------------
// argmax (main.zig:715-726) + the loop's hand-over (main.zig:999-1003, :1036)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void argmax_kernel(const ArgmaxArgs a)
{
    __shared__ float s_val[16];
    __shared__ int s_idx[16];
    __shared__ int s_next;
    const int tid = threadIdx.x;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    if (a.part_val != nullptr) {  // per-block candidates left by the classifier launch
        for (int i = tid; i < a.n_part; i += blockDim.x) {
            const float v = a.part_val[i];
            const int id = a.part_idx[i];
            if (id != 0x7fffffff && (bi == 0x7fffffff || v > best || (v == best && id < bi))) {
                best = v;
                bi = id;
            }
        }
    } else {
        for (int i = tid; i < a.vocab; i += blockDim.x) {
            const float v = a.logits[i];
            if (v > best || bi == 0x7fffffff) {  // strict '>' keeps the lowest index (:720)
                best = v;
                bi = i;
            }
        }
    }
    // wave reduce: larger value wins, equal values -> lower index
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > best || (ov == best && oi < bi))) {
            best = ov;
            bi = oi;
        }
    }
    if ((tid & 63) == 0) {
        s_val[tid >> 6] = best;
        s_idx[tid >> 6] = bi;
    }
    __syncthreads();
    if (tid == 0) {
        const int nw = blockDim.x >> 6;
        for (int w = 1; w < nw; w++) {
            if (s_idx[w] != 0x7fffffff &&
                (bi == 0x7fffffff || s_val[w] > best || (s_val[w] == best && s_idx[w] < bi))) {
                best = s_val[w];
                bi = s_idx[w];
            }
        }
        if (bi == 0x7fffffff) bi = 0;
        if (a.argmax_out) *a.argmax_out = bi;
        int next = bi;
        if (a.advance) {
            const int pos = *a.pos_ptr;
            if (pos < *a.n_prompt_ptr) next = a.prompt[pos];  // :999-1000
            a.out_tokens[pos] = next;
            *a.token_ptr = next;                              // :1036
            *a.pos_ptr = pos + 1;                             // :995
        }
        s_next = next;
    }
    __syncthreads();
    if (a.advance) {
        // next step's embedding row -> x (main.zig:295-296), saves a launch
        const float *row = a.tok_emb + (size_t)s_next * (size_t)a.dim;
        for (int i = tid; i < a.dim; i += blockDim.x) a.x[i] = row[i];
    }
}

// token/pos from the host + embedding copy (main.zig:295-296)
__global__ void set_state_kernel(int token, int pos, int *token_ptr, int *pos_ptr,
                                 const float *tok_emb, float *x, int dim)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *token_ptr = token;
        *pos_ptr = pos;
    }
    const float *row = tok_emb + (size_t)token * (size_t)dim;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < dim; i += gridDim.x * blockDim.x)
        x[i] = row[i];
}

// ---------------------------------------------------------------------------
// Stand-alone wrappers for the test hooks: same device functions as above.
// ---------------------------------------------------------------------------
// VEC: the vector staging path the fused mat-vec uses (XC = 4, zero padded)
template <bool VEC>
__global__ __launch_bounds__(kBlock) void rmsnorm_kernel(float *o, const float *x, const float *w,
                                                         int n)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (VEC) {
        const int n4 = n >> 2, n4_pad = (n4 + 255) & ~255;
        float *xs = lds, *scratch = lds + 4 * n4_pad;
        v4f xr[4], gr[4];
        xload_issue<PRO_RMS, 4>(x, w, n4, xr, gr);
        xstage_finish<PRO_RMS, 4>(x, w, n, n4_pad, xr, gr, xs, scratch);
        for (int j = threadIdx.x; j < n; j += blockDim.x) o[j] = xs[j];
    } else {
        float *xs = lds, *scratch = lds + ((n + 3) & ~3);
        stage_x_scalar<PRO_RMS>(x, w, n, xs, scratch);
        for (int j = threadIdx.x; j < n; j += blockDim.x) o[j] = xs[j];
    }
}

__global__ __launch_bounds__(kBlock) void softmax_kernel(float *x, int n)
{
    __shared__ float scratch[kScratch];
    block_softmax(x, n, scratch);
}

template <bool VEC>
__global__ __launch_bounds__(kBlock) void dot_kernel(float *out, const float *x, const float *y,
                                                     int n)
{
    // one "timestep" with head_size = n and divisor 1 (x/1 is exact)
    __shared__ float r;
    attn_scores<VEC>(x, y, 0, n, 1, 1.0f, &r);
    __syncthreads();
    if (threadIdx.x == 0) *out = r;
}

template <bool VEC>
__global__ __launch_bounds__(kBlock) void wsum_rows_kernel(float *xout, int xout_len,
                                                           const float *rows, int row_stride,
                                                           const float *weights, int n_weights,
                                                           float *part)
{
    attn_weighted_sum<VEC>(weights, rows, row_stride, xout_len, n_weights, part, xout);
}

// Seeded synthetic weights: value(idx) = bias + scale*r(idx,seed); must match
// oracle/llama2_oracle.c orc_synth_value and checkpoint.py synth_values bit for bit.
__global__ void synth_fill_kernel(float *dst, uint64_t base_idx, uint64_t count, uint64_t seed,
                                  float scale, float bias)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        uint64_t z = (base_idx + i) + seed * 0x9E3779B97F4A7C15ULL;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z = z ^ (z >> 31);
        const uint32_t u = (uint32_t)(z >> 41);
        const float r = __fsub_rn(__fmul_rn((float)u, 0x1p-22f), 1.0f);
        dst[i] = __fadd_rn(bias, __fmul_rn(scale, r));
    }
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <typename K>
hipError_t ensure_lds(K kernel, size_t bytes)
{
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
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

template <int PRO, int EPI, int LPR, int XC>
MvLaunch mv_entry()
{
    return {reinterpret_cast<const void *>(&matvec_kernel<PRO, EPI, LPR, XC>), LPR, MvGeom<LPR>::U};
}

template <int PRO, int EPI>
MvLaunch mv_pick(int lpr, bool big_x)
{
    if (lpr == 8) return mv_entry<PRO, EPI, 8, 4>();
    if (lpr == 16) return mv_entry<PRO, EPI, 16, 4>();
    if (lpr == 32) return mv_entry<PRO, EPI, 32, 4>();
    return big_x ? mv_entry<PRO, EPI, 64, 12>() : mv_entry<PRO, EPI, 64, 4>();
}

template <int PRO, int EPI>
const void *mv_row_fn(bool big_x)
{
    return big_x ? reinterpret_cast<const void *>(&matvec_row_kernel<PRO, EPI, 12>)
                 : reinterpret_cast<const void *>(&matvec_row_kernel<PRO, EPI, 4>);
}

const void *mv_row_pick(int pro, int epi, bool big_x)
{
#define L2Z_MVR(P, E) if (pro == P && epi == E) return mv_row_fn<P, E>(big_x);
    L2Z_MVR(PRO_NONE, EPI_STORE)
    L2Z_MVR(PRO_NONE, EPI_RESID)
    L2Z_MVR(PRO_RMS, EPI_STORE)
    L2Z_MVR(PRO_RMS, EPI_ROPE)
    L2Z_MVR(PRO_RMS, EPI_SWIGLU)
    L2Z_MVR(PRO_RMS, EPI_ARGMAX)
#undef L2Z_MVR
    return nullptr;
}

MvLaunch mv_pick_pe(int pro, int epi, int lpr, bool big_x, bool vec)
{
#define L2Z_MV(P, E)                                                                      \
    if (pro == P && epi == E)                                                             \
        return vec ? mv_pick<P, E>(lpr, big_x)                                            \
                   : MvLaunch{reinterpret_cast<const void *>(&matvec_scalar_kernel<P, E>), 0, 0};
    L2Z_MV(PRO_NONE, EPI_STORE)
    L2Z_MV(PRO_NONE, EPI_RESID)
    L2Z_MV(PRO_RMS, EPI_STORE)
    L2Z_MV(PRO_RMS, EPI_ROPE)
    L2Z_MV(PRO_RMS, EPI_SWIGLU)
#undef L2Z_MV
    if (pro == PRO_RMS && epi == EPI_ARGMAX && vec) return mv_pick<PRO_RMS, EPI_ARGMAX>(lpr, big_x);
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

size_t attention_lds_bytes(int head_size, int seq_len, bool vec)
{
    const AttnGeom ge = attn_geom(head_size, vec);
    size_t fl = (size_t)((head_size + 3) & ~3) + ((seq_len + 3) & ~3) + (size_t)ge.G * head_size + kScratch;
    if (vec && head_size <= 256) {  // fast kernel geometry
        const AttnGeom gf = attn_geom(head_size, true, kAttnFastBlock);
        const size_t f2 = 2 * (size_t)((seq_len + 3) & ~3) + (size_t)gf.G * head_size;
        if (f2 > fl) fl = f2;
    }
    return fl * sizeof(float);
}

int matvec_max_grid(int n_cus) { return n_cus * 8; }

// widths the vector kernels take (16-byte aligned operands assumed); the rest goes to the generic
// scalar kernel, which has no fused-argmax epilogue
bool matvec_vector_width(int n)
{
    if (n <= 0 || (n % 4) != 0) return false;
    const int n4 = n >> 2;
    return !(lpr_for(n4) == 64 && (n4 % 64) != 0);
}

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
    if (lpr == 64 && (n4 % 64) != 0) vec = false;  // rare odd widths: generic scalar kernel
    if (epi == EPI_ARGMAX && (!vec || a.rows1 != 0 || a.rows2 != 0)) return hipErrorNotSupported;
    static const int row_mode = getenv("L2Z_ROW_KERNEL") ? atoi(getenv("L2Z_ROW_KERNEL")) : 1;
    const bool use_row = vec && row_mode && n4 >= 1024 && (n4 % 64) == 0;
    MvLaunch k = mv_pick_pe(pro, epi, lpr, a.n > 4096, vec);
    if (use_row) k.fn = mv_row_pick(pro, epi, a.n > 4096);
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
    // choice is made on the whole-token rate).  L2Z_ROW_BLOCKS overrides.
    static const int row_blocks = getenv("L2Z_ROW_BLOCKS") ? atoi(getenv("L2Z_ROW_BLOCKS")) : 2;
    if (use_row && occ > row_blocks) occ = row_blocks;
    const int resident = occ * n_cus;
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
    // only the row kernel's single-segment epilogues push (wo, ffn13, ffn2, classifier)
    if (!use_row || epi == EPI_ROPE || a.rows2 != 0 || (epi != EPI_SWIGLU && a.rows1 != 0)) a.push = nullptr;
    if (pushed) *pushed = a.push != nullptr;
    void *args[] = {&a};
    return hipLaunchKernel(k.fn, dim3(grid), dim3(kBlock), args, lds, st);
}

int attention_split_chunks(int n_heads_local, int n_cus)
{
    int nch = n_cus / (n_heads_local > 0 ? n_heads_local : 1);
    if (nch > 16) nch = 16;
    if (nch < 1) nch = 1;
    return nch;
}

size_t attention_split_part_floats(int n_heads_local, int head_size, int nch)
{
    return (size_t)n_heads_local * nch * (size_t)(head_size + 4);
}

hipError_t launch_attention_split(const AttnArgs &a, int n_heads_local, int nch, float *part,
                                  hipStream_t st)
{
    const AttnGeom ge = attn_geom(a.head_size, true, kBlock);
    const int max_local = (a.seq_len + nch - 1) / nch;
    const size_t lds = (size_t)(2 * ((max_local + 3) & ~3) + ge.G * a.head_size) * sizeof(float);
    hipError_t e = ensure_lds(attention_split_kernel, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(attention_split_kernel, dim3(n_heads_local * nch), dim3(kBlock), lds, st, a,
                       nch, part);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    int ct = (a.head_size + 63) & ~63;
    if (ct > 1024) ct = 1024;
    hipLaunchKernelGGL(attention_combine_kernel, dim3(n_heads_local), dim3(ct), 0, st, part, nch,
                       a.head_size, a.xb, a.push);
    return hipGetLastError();
}

bool attention_push_supported(const AttnArgs &a)
{
    return (a.head_size % 4) == 0 && (a.kv_dim % 4) == 0 && a.head_size <= 256 && aligned16(a.q) &&
           aligned16(a.kcache) && aligned16(a.vcache);  // the fast / split kernels, not the generic one
}

bool attention_split_supported(const AttnArgs &a)
{
    return (a.head_size % 4) == 0 && a.head_size <= 256 && (a.kv_dim % 4) == 0 && aligned16(a.q) &&
           aligned16(a.kcache) && aligned16(a.vcache);
}

hipError_t launch_attention(const AttnArgs &a, int n_heads_local, hipStream_t st)
{
    const bool vec = (a.head_size % 4) == 0 && (a.kv_dim % 4) == 0 && aligned16(a.q) &&
                     aligned16(a.kcache) && aligned16(a.vcache);
    const size_t lds = attention_lds_bytes(a.head_size, a.seq_len, vec);
    if (vec && a.head_size <= 256) {
        static const int forced = getenv("L2Z_ATTN_BLOCK") ? atoi(getenv("L2Z_ATTN_BLOCK")) : 0;
        const int nt = forced ? forced : (a.seq_len > 512 ? kAttnFastBlock : kBlock);
        const AttnGeom gf = attn_geom(a.head_size, true, nt);
        const size_t lds_fast = (size_t)(2 * ((a.seq_len + 3) & ~3) + gf.G * a.head_size) * sizeof(float);
        if (nt == kAttnFastBlock) {
            hipError_t e = ensure_lds(attention_fast_kernel<kAttnFastBlock, false>, lds_fast);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((attention_fast_kernel<kAttnFastBlock, false>), dim3(n_heads_local),
                               dim3(kAttnFastBlock), lds_fast, st, a);
        } else {
            hipError_t e = ensure_lds(attention_fast_kernel<kBlock, true>, lds_fast);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((attention_fast_kernel<kBlock, true>), dim3(n_heads_local),
                               dim3(kBlock), lds_fast, st, a);
        }
        return hipGetLastError();
    }
    if (vec) {
        hipError_t e = ensure_lds(attention_kernel<true>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((attention_kernel<true>), dim3(n_heads_local), dim3(kBlock), lds, st, a);
    } else {
        hipError_t e = ensure_lds(attention_kernel<false>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((attention_kernel<false>), dim3(n_heads_local), dim3(kBlock), lds, st, a);
    }
    return hipGetLastError();
}

hipError_t launch_argmax(const ArgmaxArgs &a, hipStream_t st)
{
    hipLaunchKernelGGL(argmax_kernel, dim3(1), dim3(1024), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_set_state(int token, int pos, int *token_ptr, int *pos_ptr, const float *tok_emb,
                            float *x, int dim, hipStream_t st)
{
    const int grid = (dim + 255) / 256 > 64 ? 64 : (dim + 255) / 256;
    hipLaunchKernelGGL(set_state_kernel, dim3(grid), dim3(256), 0, st, token, pos, token_ptr,
                       pos_ptr, tok_emb, x, dim);
    return hipGetLastError();
}

hipError_t launch_rmsnorm(float *o, const float *x, const float *w, int n, hipStream_t st)
{
    const bool vec = (n % 4) == 0 && aligned16(x) && aligned16(w);
    const size_t lds = matvec_lds_bytes(n);
    if (vec) {
        hipError_t e = ensure_lds(rmsnorm_kernel<true>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((rmsnorm_kernel<true>), dim3(1), dim3(kBlock), lds, st, o, x, w, n);
    } else {
        hipError_t e = ensure_lds(rmsnorm_kernel<false>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((rmsnorm_kernel<false>), dim3(1), dim3(kBlock), lds, st, o, x, w, n);
    }
    return hipGetLastError();
}

hipError_t launch_softmax(float *x, int n, hipStream_t st)
{
    hipLaunchKernelGGL(softmax_kernel, dim3(1), dim3(kBlock), 0, st, x, n);
    return hipGetLastError();
}

hipError_t launch_dot(float *out, const float *x, const float *y, int n, hipStream_t st)
{
    const bool vec = (n % 4) == 0 && aligned16(x) && aligned16(y);
    if (vec)
        hipLaunchKernelGGL((dot_kernel<true>), dim3(1), dim3(kBlock), 0, st, out, x, y, n);
    else
        hipLaunchKernelGGL((dot_kernel<false>), dim3(1), dim3(kBlock), 0, st, out, x, y, n);
    return hipGetLastError();
}

hipError_t launch_weighted_sum_rows(float *xout, int xout_len, const float *rows, int row_stride,
                                    const float *weights, int n_weights, hipStream_t st)
{
    const bool vec = (xout_len % 4) == 0 && (row_stride % 4) == 0 && aligned16(rows) &&
                     aligned16(xout);
    float *part = nullptr;
    hipError_t e = hipMalloc(&part, (size_t)kBlock * xout_len * sizeof(float));
    if (e != hipSuccess) return e;
    if (vec)
        hipLaunchKernelGGL((wsum_rows_kernel<true>), dim3(1), dim3(kBlock), 0, st, xout, xout_len,
                           rows, row_stride, weights, n_weights, part);
    else
        hipLaunchKernelGGL((wsum_rows_kernel<false>), dim3(1), dim3(kBlock), 0, st, xout, xout_len,
                           rows, row_stride, weights, n_weights, part);
    e = hipGetLastError();
    hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(part);
    return e != hipSuccess ? e : e2;
}

hipError_t launch_synth_fill(float *dst, uint64_t base_idx, uint64_t count, uint64_t seed,
                             float scale, float bias, hipStream_t st)
{
    if (count == 0) return hipSuccess;
    uint64_t blocks = (count + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(synth_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dst, base_idx,
                       count, seed, scale, bias);
    return hipGetLastError();
}

#ifdef L2Z_DBG_TS
hipError_t dbg_ts_read(long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg_ts), 16 * sizeof(long long));
}
#endif

}  // namespace l2z
