// attention.hip -- decode attention for gfx950 (main.zig:361-389): one block per head (fast and generic
// forms) and the split / combine (flash-decoding) form.  See kernels' comments and DESIGN.md 4.2.
#include "kernel_common.h"

namespace l2z {
namespace {

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

// out[i] = part[0][i] + part[1][i] + ... + part[G-1][i], i < hs.  R = 1..16 adjacent lanes
// share one output: lane r adds partials r, r+R, ... (increasing), then a DPP sum over the R
// lanes.  R depends only on (G, hs, blockDim) -- fixed per model.
__device__ __forceinline__ void reduce_partials(const float *part, int G, int hs, float *out,
                                                const P2pArgs *push = nullptr, int push_e = 0,
                                                size_t push_idx0 = 0)
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
        if (push) p2p_ll_push(push, push_e, push_idx0 + (size_t)i, s);
    }
}

// the same sum, stored write-through (one agent-scope store per value): a hand-off to another block
__device__ __forceinline__ void reduce_partials_wt(const float *part, int G, int hs, float *out)
{
    int R = 1;
    while (R * 2 <= G && R * 2 * hs <= (int)blockDim.x && R < 16) R <<= 1;
    const int i = threadIdx.x / R, r = threadIdx.x % R;
    float s = 0.0f;
    if (i < hs)
        for (int gg = r; gg < G; gg += R) s += part[(size_t)gg * hs + i];
    s = lanes_sum(s, R);
    if (i < hs && r == 0) __hip_atomic_store(out + i, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    const float *kbase = a.kcache + (size_t)kvh * a.kv_head;  // head-major cache: this head's rows are contiguous
    const float *vbase = a.vcache + (size_t)kvh * a.kv_head;
    const size_t stride = (size_t)a.kv_row;
    const int g = threadIdx.x / ge.TPR, c0 = threadIdx.x % ge.TPR;
    const bool active = c0 < ge.E;
    const int cc = active ? c0 : 0;
    const int step = ge.G * kFastUB;
    const v4f zero = {0.f, 0.f, 0.f, 0.f};

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
    __syncthreads();
    wave_softmax(att, prob, T);  // :378
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
    if (active) ((v4f *)(part + (size_t)g * hs))[cc] = acc;
    __syncthreads();
    // sharded: a.xb already points at this rank's slice, head h of it starts at h * hs
    reduce_partials(part, ge.G, hs, a.xb + (size_t)h * hs, a.push,
                    a.push ? a.push_ctl[kCtlEpoch] + a.push_gi : 0,
                    a.push ? (size_t)a.push->rank * a.push->count + (size_t)h * hs : 0);
}

// ---------------------------------------------------------------------------
// Split attention (flash-decoding form).  One CU pulls only ~45 GB/s, so one block per
// head (attention_fast_kernel) leaves 7/8 of a 256-CU chip idle at 32 heads and spends its
// time waiting for its own K/V rows (measured with s_memtime: 6 of 10 us).  Here head h is
// shared by `nch` blocks; block (h, c) owns a contiguous range of timesteps (see the kernel) and writes
//     m_c = max score,  l_c = sum exp(score - m_c),  o_c[i] = sum exp(score - m_c) * V[t][i]
// the combine then forms  out[i] = (sum_c o_c[i] e^(m_c-M)) / (sum_c l_c e^(m_c-M)),
// M = max_c m_c.  Mathematically main.zig:361-389; in floating point the weights are
// e^(s-m_c) * e^(m_c-M) / L instead of e^(s-M) / L (a few ulp), well inside the logit
// tolerance, and independent of GPU count (attention is head-local).
// part layout: [head][chunk][head_size + 4] floats = o_c[head_size], m_c, l_c, pad, pad
// ---------------------------------------------------------------------------
// timesteps per chunk when T timesteps are cut into nch contiguous ranges: even, so that a wave's two rows stay 1 KB
// aligned.  The kernel's ranges (T = pos + 1) and the launcher's LDS carve (T = seq_len) both come from here.
__host__ __device__ inline int attn_split_per(int T, int nch) { return (((T + nch - 1) / nch) + 1) & ~1; }

// Combine of one head's chunk partials (see above); nt threads of one block, i < hs.
__device__ __forceinline__ void combine_chunks(const float *p, int nch, int hs, float *xb_h,
                                               const P2pArgs *push, int push_e, size_t push_idx0)
{
    constexpr int kMaxCh = 16;
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
        xb_h[i] = v;
        if (push) p2p_ll_push(push, push_e, push_idx0 + (size_t)i, v);
    }
}

// One launch: block (h, c) leaves its partial in global memory with write-through stores, drains
// them, and bumps head h's arrival counter; the block that arrives last (whichever it is) drops its
// stale cache lines once and combines the nch partials (the hand-off recipe of the CDNA4 guide:
// write-through payload -> drain -> counter, consumer one agent-scope acquire -> plain loads).
// No block ever waits for another.  The counter is left at 0 for the next launch.
// NT = 1024 for long contexts: 32 groups of lanes (head_size 128) x 8 rows each put a whole
// 256-timestep chunk -- K rows and V rows -- in flight in ONE round trip; with 256 threads the same
// chunk took four dependent rounds of K loads and four of V (measured 21 us per layer at pos 2047
// for 67 MB: latency-, not bandwidth-bound).
#ifdef L2Z_TIMELINE
// measurement build (scripts/timeline_build.sh): wall-clock stamps of every block of the split kernel, kept in registers
// and stored at the block's end.  Per (launch, block): [0] entry, [1] K / V loads issued, [2] scores in LDS, [3] weights in
// LDS, [4] partial reduced, [5] partial drained + arrival counted, [6] combined (the last arriver only), [7] 1 + last
constexpr int kAtlMax = 512, kAtlBlocks = 512;
__device__ long long g_atl[kAtlMax * kAtlBlocks * 8];
#define L2Z_ATL(i) tl[i] = wall_clock64()
#else
#define L2Z_ATL(i) do { } while (0)
#endif

template <int NT>
__global__ __launch_bounds__(NT) void attention_split_kernel(const AttnArgs a, int nch,
                                                             float *__restrict__ part_out,
                                                             int *__restrict__ arrivals)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef L2Z_TIMELINE
    long long tl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    L2Z_ATL(0);
    const int hs = a.head_size;
    const AttnGeom ge = attn_geom(hs, true, NT);
    const int max_local = attn_split_per(a.seq_len, nch);
    float *sc = lds;                                   // local scores
    float *wt = sc + ((max_local + 3) & ~3);           // local unnormalised weights
    float *part = wt + ((max_local + 3) & ~3);         // G*hs
    const int h = blockIdx.x / nch, c = blockIdx.x % nch;
    const int kvh = h / a.kv_mul;                      // :369
    const size_t stride = (size_t)a.kv_row;
    const int g = threadIdx.x / ge.TPR, c0 = threadIdx.x % ge.TPR;
    const bool active = c0 < ge.E;
    const int cc = active ? c0 : 0;
    const int step = ge.G * kFastUB;
    const v4f zero = {0.f, 0.f, 0.f, 0.f};

    // (pos as a kernel ARGUMENT instead of this dependent read -- what a per-replay hipGraphExecKernelNodeSetParams would
    // buy -- measured 19.68 -> 19.48 us per layer at pos 2047, 14.69 -> 14.54 at 1023: 0.13 % of a token; not built.
    // profiles/r06_attn_pos_arg.txt)
    const int T = *a.pos_ptr + 1;                      // :367
    // chunk c owns the CONTIGUOUS timesteps [c * per, c * per + Tc): in the head-major cache that is one
    // run of Tc * head_size floats of K and one of V -- a linear stream per block (the (seq_len, kv_dim)
    // order gave 512-byte pieces 16 KB apart at the 7B shape).  per is even: a wave's two rows stay
    // 1 KB aligned.  Late chunks are empty while pos is small.
    const int per = attn_split_per(T, nch);
    const int t_lo = c * per;
    const int Tc = T > t_lo ? (T - t_lo < per ? T - t_lo : per) : 0;  // timesteps owned by this block
    const float *kbase = a.kcache + (size_t)kvh * a.kv_head + (size_t)t_lo * stride;
    const float *vbase = a.vcache + (size_t)kvh * a.kv_head + (size_t)t_lo * stride;
    float *po = part_out + ((size_t)h * nch + c) * (size_t)(hs + 4);
    float m = -INFINITY, l = 0.0f;
    if (Tc == 0) {  // pos < t_lo: empty chunk (uniform branch)
        for (int i = threadIdx.x; i < hs; i += blockDim.x)
            __hip_atomic_store(po + i, 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        const v4f qv = active ? ((const v4f *)(a.q + (size_t)h * hs))[cc] : zero;
        v4f kr[kFastUB], vr[kFastUB];
        // K and V rows are read once per token and the long-context cache does not fit the on-die
        // caches: non-temporal, like the weight stream
#pragma unroll
        for (int i = 0; i < kFastUB; i++) {  // rows clamped to the chunk's last one (duplicates hit L1)
            int j = g + ge.G * i;
            j = j < Tc ? j : Tc - 1;
            kr[i] = ldg_nt((const v4f *)(kbase + (size_t)j * stride) + cc);
        }
#pragma unroll
        for (int i = 0; i < kFastUB; i++) {
            int j = g + ge.G * i;
            j = j < Tc ? j : Tc - 1;
            vr[i] = ldg_nt((const v4f *)(vbase + (size_t)j * stride) + cc);
        }
        L2Z_ATL(1);
        const float div = sqrtf((float)hs);
        for (int j0 = g;;) {  // scores (:367-375), local index j <-> t = t_lo + j
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
                kr[i] = ldg_nt((const v4f *)(kbase + (size_t)j * stride) + cc);
            }
        }
        __syncthreads();
        L2Z_ATL(2);
        // chunk-local max and sum of exponentials, redundantly per wave (identical in every wave)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int j = lane; j < Tc; j += kWave) m = fmaxf(m, sc[j]);
        m = wave_max(m);
        for (int j = lane; j < Tc; j += kWave) l += expf(sc[j] - m);
        l = wave_sum(l);
        for (int j = wave * kWave + lane; j < Tc; j += NT) wt[j] = expf(sc[j] - m);  // unnormalised
        __syncthreads();
        L2Z_ATL(3);
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
                vr[i] = ldg_nt((const v4f *)(vbase + (size_t)j * stride) + cc);
            }
        }
        if (active) ((v4f *)(part + (size_t)g * hs))[cc] = acc;
        __syncthreads();
        reduce_partials_wt(part, ge.G, hs, po);
    }
    L2Z_ATL(4);
    if (threadIdx.x == 0) {
        __hip_atomic_store(po + hs, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(po + hs + 1, l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its partial
    __syncthreads();
    int *flag = (int *)sc;  // scores are dead: reuse their first word
    if (threadIdx.x == 0) {
        const int prev = __hip_atomic_fetch_add(arrivals + h, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = prev == nch - 1;
        if (last) {
            __hip_atomic_store(arrivals + h, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // this CU's stale lines of the partials
        }
        *flag = last;
    }
    __syncthreads();
    L2Z_ATL(5);
#ifdef L2Z_TIMELINE
    const bool tl_last = *flag != 0;
    if (tl_last)
#else
    if (!*flag) return;
#endif
    combine_chunks(part_out + (size_t)h * nch * (size_t)(hs + 4), nch, hs, a.xb + (size_t)h * hs, a.push,
                   a.push ? a.push_ctl[kCtlEpoch] + a.push_gi : 0,
                   a.push ? (size_t)a.push->rank * a.push->count + (size_t)h * hs : 0);
#ifdef L2Z_TIMELINE
    if (tl_last) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        L2Z_ATL(6);
    }
    tl[7] = 1 + (tl_last ? 1 : 0);
    if (threadIdx.x == 0 && (unsigned)a.tl_seq < (unsigned)kAtlMax && blockIdx.x < kAtlBlocks) {
        long long *o = g_atl + ((size_t)a.tl_seq * kAtlBlocks + blockIdx.x) * 8;
        for (int i = 0; i < 8; i++) o[i] = tl[i];
    }
#endif
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
    attn_scores<VEC>(qs, a.kcache + (size_t)kvh * a.kv_head, a.kv_row, hs, T, sqrtf((float)hs), att);
    __syncthreads();
    block_softmax(att, T, scratch);                   // :378
    attn_weighted_sum<VEC>(att, a.vcache + (size_t)kvh * a.kv_head, a.kv_row, hs, T, part,
                           a.xb + (size_t)h * hs);    // :381-388
}

// Stand-alone wrappers for the test hooks: the same device functions as the attention kernels.
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


}  // namespace

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

// Below this position the one-block-per-head kernel runs in its 256-thread form with the speculative first
// round whatever the model's seq_len (0: never): a short context is a latency chain -- pos, q, K and V in ONE
// round trip, 4 waves instead of 16 to launch, 16 groups instead of 64 to combine -- and the 1024-thread
// form only pays once a head has more rows than 256 threads keep in flight.  Two rounds of the 256-thread
// geometry: 256 positions at head sizes <= 64 (stories110M: 7.3 -> ~4.5 us per layer at pos < 66), 128 at
// head size 128 (7B: the forms cross there, profiles/r02_kind_scan.txt).  A function of the MODEL only, so
// every rank of a shard group takes the same form at the same position.
int attention_short_pos(int head_size, int seq_len)
{
    if (seq_len <= 512 || (head_size % 4) != 0 || head_size > 256) return 0;  // that form at every position anyway
    const AttnGeom ge = attn_geom(head_size, true, kBlock);
    const int two_rounds = 2 * ge.G * kFastUB;
    return two_rounds < 256 ? two_rounds : 256;
}

// Chunks per head of the split form: one block per CU, at most 8.  More chunks than that shorten nothing -- the
// launch is a latency chain (partials, counter, combine), not a stream: stories110M (12 heads), us per layer at
// pos 256 / 1023: 16 chunks 9.3 / 9.8, 8 chunks 8.3 / 9.0, 4 chunks 8.0 / 11.8 (profiles/r03_attn_split_scan_110M.txt).
int attention_split_chunks(int n_heads_local, int n_cus)
{
    int nch = n_cus / (n_heads_local > 0 ? n_heads_local : 1);
    if (nch > 8) nch = 8;
    if (nch < 1) nch = 1;
    return nch;
}

size_t attention_split_part_floats(int n_heads_local, int head_size, int nch)
{
    return (size_t)n_heads_local * nch * (size_t)(head_size + 4);
}

// Positions from which the split form runs 1024 threads per block; below, 256.  With 8 chunks per head (7B,
// profiles/r03_attn_split_scan.txt, us per layer back to back): pos 256 10.0 vs 12.4, pos 511 11.1 vs 13.2, pos 1023
// 14.9 vs 15.0, pos 2047 21.8 vs 19.9 -- a chunk of <= 128 timesteps is four rounds of a 256-thread block's rows and
// a quarter of the waves to launch and to combine.  A function of the model and the position only.
int attention_split_wide_pos(int seq_len)
{
    return seq_len > 512 ? 1024 : seq_len;  // small contexts: 256 threads throughout (as before)
}

// small: 256 threads per block (the position is below attention_split_wide_pos)
hipError_t launch_attention_split(const AttnArgs &a_in, int n_heads_local, int nch, float *part,
                                  int *arrivals, hipStream_t st, bool small)
{
    const AttnArgs &a = a_in;
    const int nt = small ? kBlock : kAttnFastBlock;
    const AttnGeom ge = attn_geom(a.head_size, true, nt);
    const int max_local = attn_split_per(a.seq_len, nch);  // the kernel's own bound on a chunk's length
    const size_t lds = (size_t)(2 * ((max_local + 3) & ~3) + ge.G * a.head_size) * sizeof(float);
    if (nt == kAttnFastBlock) {
        hipError_t e = ensure_lds(attention_split_kernel<kAttnFastBlock>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(attention_split_kernel<kAttnFastBlock>, dim3(n_heads_local * nch),
                           dim3(kAttnFastBlock), lds, st, a, nch, part, arrivals);
    } else {
        hipError_t e = ensure_lds(attention_split_kernel<kBlock>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(attention_split_kernel<kBlock>, dim3(n_heads_local * nch), dim3(kBlock), lds,
                           st, a, nch, part, arrivals);
    }
    return hipGetLastError();
}

bool attention_push_supported(const AttnArgs &a)
{
    return (a.head_size % 4) == 0 && (a.kv_row % 4) == 0 && (a.kv_head % 4) == 0 && a.head_size <= 256 && aligned16(a.q) &&
           aligned16(a.kcache) && aligned16(a.vcache);  // the fast / split kernels, not the generic one
}

bool attention_split_supported(const AttnArgs &a)
{
    return (a.head_size % 4) == 0 && a.head_size <= 256 && (a.kv_row % 4) == 0 && (a.kv_head % 4) == 0 && aligned16(a.q) &&
           aligned16(a.kcache) && aligned16(a.vcache);
}

// form: 0 = by shape (1024 threads per head for seq_len > 512, else 256 speculative), 1 / 2 force the
// 256- / 1024-thread fast kernel, 4 forces the generic kernel (tests drive every form directly)
hipError_t launch_attention(const AttnArgs &a_in, int n_heads_local, hipStream_t st, int form)
{
    const AttnArgs &a = a_in;
    const bool vec = (a.head_size % 4) == 0 && (a.kv_row % 4) == 0 && (a.kv_head % 4) == 0 && aligned16(a.q) &&
                     aligned16(a.kcache) && aligned16(a.vcache);
    const size_t lds = attention_lds_bytes(a.head_size, a.seq_len, vec);
    if (vec && a.head_size <= 256 && form != 4) {
        const int forced = form == 1 ? kBlock : form == 2 ? kAttnFastBlock : 0;
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


}  // namespace l2z

#ifdef L2Z_TIMELINE
extern "C" int l2z_attn_timeline_dump(long long *out, int max_launches)
{
    const int n = max_launches < l2z::kAtlMax ? max_launches : l2z::kAtlMax;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(l2z::g_atl), (size_t)n * l2z::kAtlBlocks * 8 * sizeof(long long)) != hipSuccess) return 1;
    return 0;
}
#endif
