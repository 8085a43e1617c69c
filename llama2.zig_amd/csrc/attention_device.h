// attention_device.h -- the one-block-per-head decode attention (main.zig:361-389) as a device function: the body of
// attention_fast_kernel (attention.hip) and of the attention tail of the fused q | k | v launch (matvec.hip,
// EPI_ROPE_ATTN).  Anonymous namespace, like kernel_common.h.
#pragma once
#include "kernel_common.h"

namespace l2z {
namespace {

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
// head h of the launch's local heads, by the NT threads of one block; lds: 2 * round4(lds_seq) + G * hs floats, lds_seq >
// pos (the kernel: seq_len; the fused launch: the short-context bound)
template <int NT, bool SPEC>
__device__ __forceinline__ void attn_fast_head(const AttnArgs &a, int h, float *lds, int lds_seq)
{
    const int hs = a.head_size;
    const AttnGeom ge = attn_geom(hs, true, NT);
    float *att = lds;                                  // raw scores
    float *prob = att + ((lds_seq + 3) & ~3);          // softmax weights
    float *part = prob + ((lds_seq + 3) & ~3);         // G*hs
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

}  // namespace
}  // namespace l2z
