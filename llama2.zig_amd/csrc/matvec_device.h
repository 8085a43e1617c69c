// matvec_device.h -- device helpers of the mat-vec kernels (matvec.hip): the argument block copied into locals,
// the pair -> weight rows map, and the fused epilogues (RoPE + KV write, residual, SwiGLU, store) with what they
// prefetch.  Anonymous namespace, like kernel_common.h.
#pragma once
#include "kernel_common.h"

namespace l2z {
namespace {

// Kernel arguments copied into plain locals once (keeps them out of scratch).
struct MvLocals {
    const float *w0, *w1, *w2;
    float *out0, *out1, *out2;
    const float *resid;
    const float2 *rope;
    int rows0, r01, total_rows, n_pairs, n, head_size, rope_segs, pos;
    size_t ps1, ps2;
    size_t kv_head_stride;  // != 0: out1 / out2 are head-major caches (MatvecArgs::kv_head_stride)
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
    m.kv_head_stride = (EPI == EPI_ROPE) ? a.kv_head_stride : 0;
    m.push = a.push;
    m.push_e = m.push ? a.push_ctl[kCtlEpoch] + a.push_gi : 0;
    m.push_base = m.push ? (size_t)m.push->rank * m.push->count : 0;
    return m;
}

// the two weight rows of pair p (clamped to the last pair for idle lane groups)
template <int EPI>
__device__ __forceinline__ void pair_rows(const MvLocals &m, int p, const float *&pa, const float *&pb)
{
    if (p >= m.n_pairs) p = m.n_pairs - 1;
    if (EPI == EPI_SWIGLU) {  // w0: W1 | W3 row-interleaved (MatvecArgs): the pair is one contiguous run like any other
        pa = m.w0 + (size_t)(2 * p) * (size_t)m.n;
        pb = pa + m.n;
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
__device__ __forceinline__ void pair_epilogue(const MvLocals &m, int p, float sa, float sb, bool writer, const EpiIn &in)
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
            size_t ia = (size_t)row_a, ib = (size_t)row_b;
            if (m.kv_head_stride) {  // head-major cache: [kv head][pos][i]; ps1 / ps2 = pos * head_size
                const int hs = m.head_size;
                if (a1) ia = (size_t)(row_a / hs) * m.kv_head_stride + (size_t)(row_a % hs);
                if (b1) ib = (size_t)(row_b / hs) * m.kv_head_stride + (size_t)(row_b % hs);
            }
            oa[ia] = o0;
            if (valid_b) ob[ib] = o1;
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

}  // namespace
}  // namespace l2z
