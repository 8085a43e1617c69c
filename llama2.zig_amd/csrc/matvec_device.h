// matvec_device.h -- device helpers shared by the mat-vec kernels (matvec.hip) and the persistent decode engine
// (engine.hip): the argument block copied into locals, the pair -> weight rows map, and the fused epilogues
// (RoPE + KV write, residual, SwiGLU, store) with what they prefetch.  Anonymous namespace, like kernel_common.h.
#pragma once
#include <type_traits>

#include "kernel_common.h"

namespace l2z {
namespace {

// Kernel arguments copied into plain locals once (keeps them out of scratch).
// G: the pointers carry the device-memory address space explicitly (engine.hip reads them out of a structure in
// memory, where the compiler cannot infer it and would emit flat loads / stores); !G: plain pointers (kernel arguments)
template <bool G>
struct MvLocalsT {
    typedef typename std::conditional<G, const L2Z_G float *, const float *>::type CFP;
    typedef typename std::conditional<G, L2Z_G float *, float *>::type FP;
    typedef typename std::conditional<G, const L2Z_G float2 *, const float2 *>::type CF2P;
    CFP w0, w1, w2;
    FP out0, out1, out2;
    CFP resid;
    CF2P rope;
    int rows0, r01, total_rows, n_pairs, n, head_size, rope_segs, pos;
    size_t ps1, ps2;
    size_t kv_head_stride;  // != 0: out1 / out2 are head-major caches (MatvecArgs::kv_head_stride)
    const P2pArgs *push;  // sharded: LL words of the outputs go straight to the peers
    int push_e;
    size_t push_base;     // index of out0[0] in the gathered vector
    const unsigned long long *resid_slot;  // EPI_RESID, overlapped chain: residual as LL words (else null)
    unsigned resid_e;
    int *resid_ctl;
    int *resid_herr;
    long long resid_timeout;
    bool resid_pre;  // duo kernel: the residual words of all the block's units were requested at entry (EpiIn::rw is filled from LDS)
};
typedef MvLocalsT<false> MvLocals;

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
    m.resid_slot = nullptr; m.resid_e = 0; m.resid_ctl = nullptr; m.resid_herr = nullptr; m.resid_timeout = 0; m.resid_pre = false;
    if (EPI == EPI_RESID && a.resid_in.slots != nullptr) {
        const int e = a.resid_in.ctl[kCtlEpoch] + a.resid_in.gi;
        m.resid_e = (unsigned)e;
        m.resid_slot = a.resid_in.slots + (size_t)(e & 1) * a.resid_in.slot_floats;
        m.resid_ctl = a.resid_in.ctl; m.resid_herr = a.resid_in.h_err; m.resid_timeout = a.resid_in.timeout_ticks;
    }
    return m;
}

// the two weight rows of pair p (clamped to the last pair for idle lane groups)
template <int EPI, bool G>
__device__ __forceinline__ void pair_rows(const MvLocalsT<G> &m, int p, typename MvLocalsT<G>::CFP &pa,
                                          typename MvLocalsT<G>::CFP &pb)
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
        typename MvLocalsT<G>::CFP wa = a1 ? m.w1 : m.w0;
        wa = a2 ? m.w2 : wa;
        typename MvLocalsT<G>::CFP wb = b1 ? m.w1 : m.w0;
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
    v4u rw;  // LL residual: the two words as loaded at prefetch time (validated in the epilogue)
};

template <int EPI>
__device__ __forceinline__ EpiIn epi_prefetch(const MvLocals &m, int p, bool writer)
{
    EpiIn e;
    e.ra = 0.0f; e.rb = 0.0f; e.cs = make_float2(1.0f, 0.0f); e.rw = v4u{0u, 0u, 0u, 0u};
    if (!writer || p >= m.n_pairs) return e;
    if (EPI == EPI_RESID) {  // single segment: rows 2p, 2p+1
        const int ga = 2 * p, gb = ga + 1;
        if (m.resid_slot) {  // words 2p, 2p+1 of the handed-over vector: one 16-byte load, never waited for here
            if (!m.resid_pre) e.rw = ll_load2(m.resid_slot, (size_t)ga);
        } else {
            e.ra = m.resid[ga];
            if (gb < m.total_rows) e.rb = m.resid[gb];
        }
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

// stash != null (duo kernel of an overlapped chain): the values that would be pushed as LL words are left in
// stash[0], stash[1] instead and pushed by the block when its units are done (matvec_duo_kernel)
template <int EPI, bool G>
__device__ __forceinline__ void pair_epilogue(const MvLocalsT<G> &m, int p, float sa, float sb,
                                              bool writer, const EpiIn &in, float *stash = nullptr)
{
    const bool valid_a = p < m.n_pairs;
    if (EPI == EPI_SWIGLU) {
        float v = sa;
        v = v * (1.0f / (1.0f + expf(-v)));  // :412
        v = v * sb;                          // :416
        if (writer && valid_a) {
            m.out0[p] = v;
            if (stash) stash[0] = v;
            else if (m.push) p2p_ll_push(m.push, m.push_e, m.push_base + (size_t)p, v);
        }
        return;
    }
    const int ga = 2 * p, gb = ga + 1;
    const bool valid_b = valid_a && gb < m.total_rows;
    const bool a1 = ga >= m.rows0, a2 = ga >= m.r01;
    const bool b1 = gb >= m.rows0, b2 = gb >= m.r01;
    const int row_a = ga - (a2 ? m.r01 : (a1 ? m.rows0 : 0));
    const int row_b = gb - (b2 ? m.r01 : (b1 ? m.rows0 : 0));
    typename MvLocalsT<G>::FP oa = a1 ? m.out1 + m.ps1 : m.out0;
    oa = a2 ? m.out2 + m.ps2 : oa;
    typename MvLocalsT<G>::FP ob = b1 ? m.out1 + m.ps1 : m.out0;
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
            float ra = in.ra, rb = in.rb;
            if (m.resid_slot) {  // the prefetched words carry their epoch; late ones (never, in practice) are re-read
                v4u w = in.rw;
                const long long t0 = wall_clock64();
                while (!(w.y == m.resid_e && (w.w == m.resid_e || !valid_b))) {
                    if (__hip_atomic_load(m.resid_ctl + kCtlErr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                    if (wall_clock64() - t0 > m.resid_timeout) {
                        __hip_atomic_store(m.resid_ctl + kCtlErr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        *m.resid_herr = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(4);
                    w = ll_load2(m.resid_slot, (size_t)ga);
                }
                ra = __uint_as_float(w.x);
                rb = __uint_as_float(w.z);
            }
            const float va = ra + sa, vb = rb + sb;  // :711 a[i] += b[i]  (resid[row] prefetched)
            oa[row_a] = va;
            if (valid_b) ob[row_b] = vb;
            if (stash) {
                stash[0] = va;
                stash[1] = vb;
            } else if (m.push) {  // single segment on this path: row == index in the slice
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
