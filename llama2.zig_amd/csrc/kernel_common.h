// kernel_common.h -- device helpers shared by the gfx950 kernel files (matvec.hip, attention.hip,
// misc_kernels.hip): wave64 reductions on DPP, float4 arithmetic, the x staging (+ rmsnorm)
// prologue, the block softmax, launch helpers.  Everything is in an anonymous namespace: each
// translation unit gets its own copy.  Compiled with -ffp-contract=off.
#pragma once
#include <cstdlib>

#include "l2z_comm.h"
#include "l2z_internal.h"
#include "tunables.h"

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

// Tail of the staging shared by the plain and the LL form: xr[] holds this thread's first XC
// float4 of x; `loadx(j)` fetches float4 j of x for the part of a long x that is not in registers.
template <int PRO, int XC, typename LoadX>
__device__ __forceinline__ void xstage_tail(const float *__restrict__ rms_w, int n, int n4_pad,
                                            v4f (&xr)[XC], v4f (&gr)[XC], float *xs, float *scratch,
                                            LoadX loadx)
{
    const int tid = threadIdx.x;
    const int n4 = n >> 2;
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
        const v4f v = (j < n4) ? loadx(j) : v4f{0.f, 0.f, 0.f, 0.f};
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

template <int PRO, int XC>
__device__ __forceinline__ void xstage_finish(const float *__restrict__ x,
                                              const float *__restrict__ rms_w, int n, int n4_pad,
                                              v4f (&xr)[XC], v4f (&gr)[XC], float *xs,
                                              float *scratch)
{
    const v4f *x4 = (const v4f *)x;
    xstage_tail<PRO, XC>(rms_w, n, n4_pad, xr, gr, xs, scratch, [&](int j) { return x4[j]; });
}

// ---------------------------------------------------------------------------
// The same staging when x is a GATHERED vector of a sharded run (peer-write transport, p2p.hip):
// instead of a separate gather launch copying the peers' slices into a plain buffer, the consumer
// reads the LL words {value, epoch} straight out of this rank's own landing slot -- every rank,
// this one included, stored its slice there -- and re-reads until every word carries the epoch of
// the gather.  Four words (two 16-byte system-scope loads) make one float4 of x.
// ---------------------------------------------------------------------------
struct LLPoll {
    const unsigned long long *slot;  // landing slot of this gather
    unsigned e;
    unsigned count;
    int *ctl;
    int *h_err;
    long long timeout_ticks;
};

__device__ __forceinline__ LLPoll ll_poll_init(const LLIn &in)
{
    LLPoll p;
    const int e = in.ctl[kCtlEpoch] + in.gi;
    p.e = (unsigned)e;
    p.slot = in.slots + (size_t)(e & 1) * in.slot_floats;
    p.count = in.count;
    p.ctl = in.ctl;
    p.h_err = in.h_err;
    p.timeout_ticks = in.timeout_ticks;
    return p;
}

// float4 j of the gathered vector; a, b = the two loads already made (re-made until ready)
__device__ __forceinline__ v4f ll_wait4(const LLPoll &p, int j, v4u a, v4u b)
{
    if (!(ll_ready2(a, p.e) && ll_ready2(b, p.e))) {
        // slow path: a peer (or this rank's own producer on another stream: never) is behind
        bool give_up = __hip_atomic_load(p.ctl + kCtlErr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        const long long t0 = wall_clock64();
        while (!give_up) {
            __builtin_amdgcn_s_sleep(16);
            a = ll_load2(p.slot, (size_t)4 * j);
            b = ll_load2(p.slot, (size_t)4 * j + 2);
            if (ll_ready2(a, p.e) && ll_ready2(b, p.e)) break;
            if (wall_clock64() - t0 > p.timeout_ticks) {
                give_up = true;
                __hip_atomic_store(p.ctl + kCtlErr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *p.h_err = 1 + (int)((unsigned)(4 * j) / (p.count ? p.count : 1u));
            }
        }
    }
    v4f v;
    v.x = __uint_as_float(a.x);
    v.y = __uint_as_float(a.z);
    v.z = __uint_as_float(b.x);
    v.w = __uint_as_float(b.z);
    return v;
}

template <int PRO, int XC>
__device__ __forceinline__ void xload_issue_ll(const LLPoll &p, const float *__restrict__ rms_w,
                                               int n4, v4u (&xl)[2 * XC], v4f (&gr)[XC])
{
    const v4f *g4 = (const v4f *)rms_w;
#pragma unroll
    for (int k = 0; k < XC; k++) {
        const int j = threadIdx.x + kBlock * k;
        const int jc = j < n4 ? j : 0;  // past the end: any valid words, the value is dropped below
        xl[2 * k] = ll_load2(p.slot, (size_t)4 * jc);
        xl[2 * k + 1] = ll_load2(p.slot, (size_t)4 * jc + 2);
    }
    if (PRO == PRO_RMS) {
#pragma unroll
        for (int k = 0; k < XC; k++) {
            const int j = threadIdx.x + kBlock * k;
            gr[k] = (j < n4) ? g4[j] : v4f{0.f, 0.f, 0.f, 0.f};
        }
    }
}

template <int PRO, int XC>
__device__ __forceinline__ void xstage_finish_ll(const LLPoll &p, const float *__restrict__ rms_w,
                                                 int n, int n4_pad, v4u (&xl)[2 * XC],
                                                 v4f (&gr)[XC], float *xs, float *scratch)
{
    const int n4 = n >> 2;
    v4f xr[XC];
#pragma unroll
    for (int k = 0; k < XC; k++) {
        const int j = threadIdx.x + kBlock * k;
        const v4f v = ll_wait4(p, j < n4 ? j : 0, xl[2 * k], xl[2 * k + 1]);
        xr[k] = (j < n4) ? v : v4f{0.f, 0.f, 0.f, 0.f};
    }
    xstage_tail<PRO, XC>(rms_w, n, n4_pad, xr, gr, xs, scratch, [&](int j) {
        return ll_wait4(p, j, ll_load2(p.slot, (size_t)4 * j), ll_load2(p.slot, (size_t)4 * j + 2));
    });
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

}  // namespace
}  // namespace l2z
