// fused_small.hip -- small, cache-resident models (the stories15M shape): rmsnorm + q/k/v rows +
// RoPE + KV-cache write + attention of ONE head in one launch (main.zig:305-389), one block per head.
//
// Why.  These models are launch-bound, not bandwidth-bound (60 MB of weights sit in the on-die
// caches): a launch costs a 1.6 us graph-node boundary plus one dependent memory
// round trip before it can do anything.  qkv -> attention is the only seam of a layer without an
// all-to-all dependency -- head h's scores need only q_h, k_h, v_h (main.zig:361-389), which are
// rows h*hs .. (h+1)*hs of wq / wk / wv times the same normalised x -- so it is the one seam that
// can go without a device-wide hand-off.  Fused, the K and V rows of every earlier timestep are
// requested at kernel start (they do not depend on this token) as direct-to-LDS loads
// (global_load_lds_dwordx4: no VGPR is tied up while the block computes its 3*hs dot products --
// a 1024-thread block has 128 per lane, and holding the rows in registers spilled) and are read
// back from LDS by the lane that asked for them; 5 -> 4 launches per layer.
//
// One block = 1024 threads = 16 waves.  Phase 1: x -> LDS with rmsnorm (main.zig:432-468, eps after
// the divide).  Phase 2: the block's 3*hs/2 row pairs ((i, i+1) = the RoPE pair) are dealt to lane
// groups of LPR lanes (LPR from the row length as in matvec.hip; a whole row pair is in flight per
// group), rotated (main.zig:346-349) and written to q (LDS + RunState.q), to row `pos` of the K
// cache and of the V cache (main.zig:354-358) and kept in LDS for this token.  Phase 3: the
// attention of attention.hip's one-block-per-head kernel with G = 1024 / TPR groups: scores with
// the divide by sqrt(head_size) (:372), softmax (:687-706), weighted sum of V rows in increasing t
// per group (:657-685).
//
// Summation orders differ from the unfused launches (16 wave partials in the rmsnorm, G groups in
// the weighted sum): results agree within fp32 reassociation, not bit for bit.  Sharded runs and
// L2Z_FUSE_SMALL=0 use the unfused launches.
#include "kernel_common.h"

namespace l2z {
namespace {

constexpr int kFusedBlock = 1024;
constexpr int kFusedUB = 4;  // timesteps per lane group: seq_len <= 4 G (K and V rows of all of them sit in LDS)
constexpr int kFusedWaves = kFusedBlock / kWave;
constexpr int kRowU = 6;  // float4 per row per lane: LPR * 6 >= n/4 (fused_small_supported)

template <int LPR>
__global__ __launch_bounds__(kFusedBlock) void fused_qkv_attn_kernel(const FusedQkvAttnArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hs = a.head_size, n = a.n, n4 = n >> 2;
    const int n4_pad = ((n4 + LPR * kRowU - 1) / (LPR * kRowU)) * (LPR * kRowU);
    const int S4 = (a.seq_len + 3) & ~3;
    const AttnGeom ge = attn_geom(hs, true, kFusedBlock);
    float *xs = lds;                       // n4_pad * 4 (zero padded)
    float *scratch = xs + 4 * n4_pad;      // kScratch
    float *cur = scratch + kScratch;       // q_h | k_h | v_h of this token: 3 * hs
    float *att = cur + ((3 * hs + 3) & ~3);  // seq_len raw scores
    float *prob = att + S4;                // seq_len softmax weights
    float *part = prob + S4;               // G * hs
    v4f *kv_lds = (v4f *)(part + (size_t)ge.G * hs);  // [K | V][kFusedUB][1024] float4, slot [i][tid]
    constexpr int UB = kFusedUB;
    const v4f zero = {0.f, 0.f, 0.f, 0.f};
    const int h = blockIdx.x;

    // ---- requests that depend on nothing this kernel computes: x, the rmsnorm weight, and the K / V
    // rows of every timestep this block may need (rows >= pos are masked or replaced below)
    const v4f xv = tid < n4 ? ((const v4f *)a.x)[tid] : zero;       // n4 <= kFusedBlock (supported())
    const v4f gv = tid < n4 ? ((const v4f *)a.rms_w)[tid] : zero;
    const int g = tid / ge.TPR, c0 = tid % ge.TPR;
    const bool active = c0 < ge.E;
    const int cc = active ? c0 : 0;
    const size_t stride = (size_t)a.kv_row;
    const float *kbase = a.kcache + (size_t)h * a.kv_head, *vbase = a.vcache + (size_t)h * a.kv_head;
    // lane L of wave w asks for the float4 it will use itself: LDS slot [i][tid] (a wave's 64 slots
    // are the 1 KB the instruction writes, lane-linear from the wave-uniform base)
#pragma unroll
    for (int i = 0; i < UB; i++) {
        int t = g + ge.G * i;
        t = t < a.seq_len ? t : a.seq_len - 1;
        __builtin_amdgcn_global_load_lds((const v4f *)(kbase + (size_t)t * stride) + cc,
                                         kv_lds + (size_t)i * kFusedBlock + (tid & ~63), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < UB; i++) {
        int t = g + ge.G * i;
        t = t < a.seq_len ? t : a.seq_len - 1;
        __builtin_amdgcn_global_load_lds((const v4f *)(vbase + (size_t)t * stride) + cc,
                                         kv_lds + (size_t)(UB + i) * kFusedBlock + (tid & ~63), 16, 0, 0);
    }
    const int pos = *a.pos_ptr;
    const int T = pos + 1;  // timesteps 0..pos inclusive (:367)

    // ---- this wave's first unit of row pairs: weights requested before x is staged.
    // A unit = RW pairs of ONE segment (q, k or v rows of this head), so the segment -- and with it
    // the matrix -- is wave-uniform: the loads are buffer loads off a scalar descriptor with one
    // 32-bit offset register each (per-lane 64-bit addresses for 12 loads in flight spilled).
    constexpr int RW = kWave / LPR;
    const int grp = lane / LPR, cl = lane % LPR;
    const int half = hs >> 1;             // pairs per segment of this head
    const int ups = (half + RW - 1) / RW; // units per segment
    const int n_units = 3 * ups;
    const unsigned mat_bytes = (unsigned)((size_t)n * (size_t)a.kv_dim * sizeof(float));  // MHA: dim x dim
    v4f wa[kRowU], wb[kRowU];
    int seg = 0, lp = 0;
    bool valid = false;
    auto issue_unit = [&](int uu) {
        seg = __builtin_amdgcn_readfirstlane(uu / ups);
        lp = (uu - seg * ups) * RW + grp;
        valid = lp < half;
        const int lpc = valid ? lp : half - 1;
        const float *wbase = seg == 0 ? a.wq : (seg == 1 ? a.wk : a.wv);
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wbase), 0, mat_bytes, 0x00020000);
        // One offset register per row; the column steps are instruction immediates.  Columns past
        // the row end read on into the next rows (finite values; past the matrix the buffer returns
        // 0): their x entries are the zero padding, so they add exactly 0 -- no clamp, no select.
        const unsigned off_a = (unsigned)(((size_t)h * hs + 2 * lpc) * (size_t)n * sizeof(float)) + 16u * cl;
        const unsigned off_b = off_a + 4u * (unsigned)n;
#pragma unroll
        for (int k = 0; k < kRowU; k++) {  // cacheable loads: these models' weights are cache resident
            wa[k] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, off_a + 16u * LPR * k, 0, 0));
            wb[k] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, off_b + 16u * LPR * k, 0, 0));
        }
    };
    int u = wave;
    issue_unit(u < n_units ? u : 0);
    float2 cs = make_float2(1.0f, 0.0f);
    if (seg < 2) cs = a.rope[(size_t)pos * (size_t)half + (size_t)(valid ? lp : 0)];  // (i % head_size) / 2 == lp

    // ---- phase 1: rmsnorm(x) -> LDS (main.zig:432-468)
    {
        float ss = 0.0f;
        ss = fmaf(xv.x, xv.x, ss);
        ss = fmaf(xv.y, xv.y, ss);
        ss = fmaf(xv.z, xv.z, ss);
        ss = fmaf(xv.w, xv.w, ss);
        ss = wave_sum(ss);
        if (lane == 0) scratch[wave] = ss;
        __syncthreads();
        float tot = scratch[0];
#pragma unroll
        for (int i = 1; i < kFusedWaves; i++) tot += scratch[i];
        float sc = tot / (float)n;   // :452
        sc += 1e-5f;                 // :453
        sc = 1.0f / sqrtf(sc);       // :454
        v4f v;
        v.x = (xv.x * sc) * gv.x;    // :462
        v.y = (xv.y * sc) * gv.y;
        v.z = (xv.z * sc) * gv.z;
        v.w = (xv.w * sc) * gv.w;
        if (tid < n4_pad) ((v4f *)xs)[tid] = tid < n4 ? v : zero;
        __syncthreads();
    }

    // ---- phase 2: the head's q, k, v rows (main.zig:308-320) + RoPE (:336-351) + cache write (:354-358)
    const v4f *xs4 = (const v4f *)xs;
    // Most waves own one unit (stories15M: 18 units, 16 waves); a wave's later units are issued only
    // after the previous one is consumed -- prefetching them would need a second set of 48 row
    // registers, which a 1024-thread block does not have.
    for (; u < n_units; u += kFusedWaves) {
        if (u != wave) {
            __builtin_amdgcn_sched_barrier(0);
            issue_unit(u);
            cs = seg < 2 ? a.rope[(size_t)pos * (size_t)half + (size_t)(valid ? lp : 0)] : make_float2(1.0f, 0.0f);
        }
        v4f acc_a = zero, acc_b = zero;
#pragma unroll
        for (int k = 0; k < kRowU; k++) {
            const v4f x4 = xs4[cl + LPR * k];
            acc_a = fma4(wa[k], x4, acc_a);
            acc_b = fma4(wb[k], x4, acc_b);
        }
        const float sa = lanes_sum(hsum4(acc_a), LPR);
        const float sb = lanes_sum(hsum4(acc_b), LPR);
        if (cl == 0 && valid) {
            float o0 = sa, o1 = sb;
            if (seg < 2) {
                o0 = sa * cs.x - sb * cs.y;  // :348
                o1 = sa * cs.y + sb * cs.x;  // :349
            }
            cur[seg * hs + 2 * lp] = o0;
            cur[seg * hs + 2 * lp + 1] = o1;
            float *dst = seg == 0 ? a.q_out + (size_t)h * hs
                                  : (seg == 1 ? a.kcache : a.vcache) + (size_t)pos * stride + (size_t)h * a.kv_head;
            dst[2 * lp] = o0;
            dst[2 * lp + 1] = o1;
        }
    }
    __syncthreads();

    // ---- phase 3: attention of head h (main.zig:361-389); row `pos` comes from this launch
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's direct-to-LDS rows have landed
    v4f kr[UB], vr[UB];
#pragma unroll
    for (int i = 0; i < UB; i++) {
        kr[i] = kv_lds[(size_t)i * kFusedBlock + tid];
        vr[i] = kv_lds[(size_t)(UB + i) * kFusedBlock + tid];
    }
    const v4f qv = active ? ((const v4f *)cur)[cc] : zero;
    const float div = sqrtf((float)hs);
#pragma unroll
    for (int i = 0; i < UB; i++) {
        const int t = g + ge.G * i;
        if (t == pos) kr[i] = active ? ((const v4f *)(cur + hs))[cc] : zero;
        float p = hsum4(fma4(qv, kr[i], zero));
        p = lanes_sum(p, ge.TPR);
        if (c0 == 0 && t < T) att[t] = p / div;  // :372 divide
    }
    __syncthreads();
    wave_softmax(att, prob, T);  // :378
    v4f acc = zero;
#pragma unroll
    for (int i = 0; i < UB; i++) {  // att . V (:381-388), increasing t within the group
        const int t = g + ge.G * i;
        if (t == pos) vr[i] = active ? ((const v4f *)(cur + 2 * hs))[cc] : zero;
        const float w = t < T ? prob[t] : 0.0f;
        acc.x = fmaf(vr[i].x, w, acc.x);
        acc.y = fmaf(vr[i].y, w, acc.y);
        acc.z = fmaf(vr[i].z, w, acc.z);
        acc.w = fmaf(vr[i].w, w, acc.w);
    }
    if (active) ((v4f *)(part + (size_t)g * hs))[cc] = acc;
    __syncthreads();
    {   // out[i] = part[0][i] + ... + part[G-1][i]: R adjacent lanes share one output
        int R = 1;
        while (R * 2 <= ge.G && R * 2 * hs <= kFusedBlock && R < 16) R <<= 1;
        const int i = tid / R, r = tid % R;
        float s = 0.0f;
        if (i < hs)
            for (int gg = r; gg < ge.G; gg += R) s += part[(size_t)gg * hs + i];
        s = lanes_sum(s, R);
        if (i < hs && r == 0) a.xb[(size_t)h * hs + i] = s;
    }
}

int fused_lpr(int n4)
{
    int l = 8;
    while (l < 64 && l * kRowU < n4) l <<= 1;
    return l;
}

size_t fused_lds_bytes(const FusedQkvAttnArgs &a)
{
    const int n4 = a.n >> 2, lpr = fused_lpr(n4);
    const int n4_pad = ((n4 + lpr * kRowU - 1) / (lpr * kRowU)) * (lpr * kRowU);
    const AttnGeom ge = attn_geom(a.head_size, true, kFusedBlock);
    return (size_t)(4 * n4_pad + kScratch + ((3 * a.head_size + 3) & ~3) + 2 * ((a.seq_len + 3) & ~3) +
                    ge.G * a.head_size) * sizeof(float) +
           (size_t)2 * kFusedUB * kFusedBlock * sizeof(v4f);  // K and V rows
}

template <int LPR>
hipError_t launch_lpr(const FusedQkvAttnArgs &a, int n_heads, size_t lds, hipStream_t st)
{
    hipError_t e = ensure_lds(fused_qkv_attn_kernel<LPR>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((fused_qkv_attn_kernel<LPR>), dim3(n_heads), dim3(kFusedBlock), lds, st, a);
    return hipGetLastError();
}

}  // namespace

// Shapes the fused launch takes: all heads' K/V rows in one speculative round, a row pair in one
// batch per lane group, x in one float4 per thread, and little enough weight per head that
// one CU pulling it does not take longer than the launch it saves (stories15M: 166 KB per head).
bool fused_qkv_attn_supported(int dim, int n_heads, int n_kv_heads, int seq_len, int n_cus)
{
    if (n_heads != n_kv_heads || dim % n_heads != 0 || dim % 4 != 0) return false;
    const int hs = dim / n_heads, n4 = dim >> 2;
    if (hs % 4 != 0 || hs > 256 || (hs & 1)) return false;
    if (n4 > kFusedBlock || fused_lpr(n4) * kRowU < n4) return false;
    const AttnGeom ge = attn_geom(hs, true, kFusedBlock);
    if (ge.G * kFusedUB < seq_len) return false;
    if ((size_t)3 * hs * dim * sizeof(float) > ((size_t)256 << 10)) return false;
    FusedQkvAttnArgs probe = {};
    probe.n = dim; probe.head_size = hs; probe.seq_len = seq_len;
    if (fused_lds_bytes(probe) > (size_t)160 * 1024) return false;
    return n_heads <= n_cus;
}

hipError_t launch_fused_qkv_attn(const FusedQkvAttnArgs &a, int n_heads, hipStream_t st)
{
    if (!aligned16(a.x) || !aligned16(a.rms_w) || !aligned16(a.wq) || !aligned16(a.wk) || !aligned16(a.wv) ||
        !aligned16(a.kcache) || !aligned16(a.vcache))
        return hipErrorInvalidValue;
    const int lpr = fused_lpr(a.n >> 2);
    const size_t lds = fused_lds_bytes(a);
    switch (lpr) {
        case 8: return launch_lpr<8>(a, n_heads, lds, st);
        case 16: return launch_lpr<16>(a, n_heads, lds, st);
        case 32: return launch_lpr<32>(a, n_heads, lds, st);
        default: return launch_lpr<64>(a, n_heads, lds, st);
    }
}

}  // namespace l2z
