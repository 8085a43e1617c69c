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
#include "l2z_internal.h"

namespace l2z {
namespace {

constexpr int kWave = 64;
constexpr int kBlock = 256;
constexpr int kWaves = kBlock / kWave;
constexpr int kScratch = 32;  // floats of LDS scratch for block reductions

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4f ldg_nt(const v4f *p) { return __builtin_nontemporal_load(p); }

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
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
// ---------------------------------------------------------------------------
template <int PRO, bool VEC>
__device__ __forceinline__ void stage_x(const float *__restrict__ x, const float *__restrict__ rms_w,
                                        int n, float *xs, float *scratch)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    float ss = 0.0f;
    if (VEC) {
        const v4f *x4 = (const v4f *)x;
        v4f *xs4 = (v4f *)xs;
        const int n4 = n >> 2;
        for (int j = tid; j < n4; j += nt) {
            const v4f v = x4[j];
            xs4[j] = v;
            if (PRO == PRO_RMS) {
                ss = fmaf(v.x, v.x, ss);
                ss = fmaf(v.y, v.y, ss);
                ss = fmaf(v.z, v.z, ss);
                ss = fmaf(v.w, v.w, ss);
            }
        }
        if (PRO == PRO_RMS) {
            const float tot = block_sum(ss, scratch);
            float s = tot / (float)n;
            s += 1e-5f;
            s = 1.0f / sqrtf(s);
            const v4f *g4 = (const v4f *)rms_w;
            for (int j = tid; j < n4; j += nt) {  // same j this thread wrote above
                v4f v = xs4[j];
                const v4f g = g4[j];
                v.x = (v.x * s) * g.x;
                v.y = (v.y * s) * g.y;
                v.z = (v.z * s) * g.z;
                v.w = (v.w * s) * g.w;
                xs4[j] = v;
            }
        }
    } else {
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
    }
    __syncthreads();
}

// Two dot products against the staged x: rows pa and pb.  main.zig:553-604,
// summation order: lane l takes float4 columns l, l+64, ... in increasing
// order into 4 component accumulators, then (x+y)+(z+w), then xor-shuffle.
// The order depends only on n, never on the grid or on how rows are sharded.
template <bool VEC>
__device__ __forceinline__ void dot2(const float *__restrict__ pa, const float *__restrict__ pb,
                                     const float *xs, int n, float &ra, float &rb)
{
    const int lane = threadIdx.x & 63;
    if (VEC) {
        constexpr int U = 4;  // 2 rows x 4 x 16 B = 8 loads (8 KiB per wave) in flight
        const v4f *a4 = (const v4f *)pa + lane;
        const v4f *b4 = (const v4f *)pb + lane;
        const v4f *xs4 = (const v4f *)xs + lane;
        const int n4 = n >> 2;
        v4f acc_a = {0.f, 0.f, 0.f, 0.f}, acc_b = {0.f, 0.f, 0.f, 0.f};
        int j = lane;
        for (; j + kWave * (U - 1) < n4; j += kWave * U) {
            v4f wa[U], wb[U];
#pragma unroll
            for (int k = 0; k < U; k++) {  // constant 1 KiB strides -> immediate offsets
                wa[k] = ldg_nt(a4 + kWave * k);
                wb[k] = ldg_nt(b4 + kWave * k);
            }
#pragma unroll
            for (int k = 0; k < U; k++) {
                const v4f xv = xs4[kWave * k];
                acc_a = fma4(wa[k], xv, acc_a);
                acc_b = fma4(wb[k], xv, acc_b);
            }
            a4 += kWave * U;
            b4 += kWave * U;
            xs4 += kWave * U;
        }
        for (; j < n4; j += kWave) {
            const v4f wa = ldg_nt(a4), wb = ldg_nt(b4);
            const v4f xv = *xs4;
            acc_a = fma4(wa, xv, acc_a);
            acc_b = fma4(wb, xv, acc_b);
            a4 += kWave;
            b4 += kWave;
            xs4 += kWave;
        }
        ra = wave_sum(hsum4(acc_a));
        rb = wave_sum(hsum4(acc_b));
    } else {
        float sa = 0.0f, sb = 0.0f;
        for (int j = lane; j < n; j += kWave) {
            const float xv = xs[j];
            sa = fmaf(pa[j], xv, sa);
            sb = fmaf(pb[j], xv, sb);
        }
        ra = wave_sum(sa);
        rb = wave_sum(sb);
    }
}

// ---------------------------------------------------------------------------
// The fused mat-vec.  main.zig:530-605 matmul_fused with its neighbours:
//   PRO_RMS    rmsnorm before it                   :305 / :398 / :426
//   EPI_ROPE   RoPE on q,k + KV-cache row write    :336-358
//   EPI_RESID  accum into the residual stream      :395 / :422
//   EPI_SWIGLU silu(w1.x) * (w3.x)                 :411-416
// All kernel arguments are read into scalars up front and selected with plain
// arithmetic: indexing the by-value argument block dynamically would push it
// into scratch memory (136 B/lane and half the occupancy in the first build).
// ---------------------------------------------------------------------------
template <int PRO, int EPI, bool VEC>
__global__ __launch_bounds__(kBlock) void matvec_kernel(const MatvecArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int n = a.n;
    float *xs = lds;
    float *scratch = lds + ((n + 3) & ~3);
    stage_x<PRO, VEC>(a.x, a.rms_w, n, xs, scratch);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *const w0 = a.w0, *const w1 = a.w1, *const w2 = a.w2;
    float *const out0 = a.out0, *const out1 = a.out1, *const out2 = a.out2;
    const int rows0 = a.rows0, rows1 = a.rows1, rows2 = a.rows2;
    const int total_rows = rows0 + rows1 + rows2;
    const int n_units = (EPI == EPI_SWIGLU) ? rows0 : (total_rows + 1) >> 1;
    const int pos = (EPI == EPI_ROPE) ? *a.pos_ptr : 0;
    const size_t ps1 = (size_t)pos * (size_t)a.pos_stride1, ps2 = (size_t)pos * (size_t)a.pos_stride2;

    for (int u = blockIdx.x * kWaves + wave; u < n_units; u += gridDim.x * kWaves) {
        // rows ga, gb in the concatenated row space [0, rows0+rows1+rows2)
        const float *pa, *pb;
        float *oa, *ob;
        int row_a, row_b, seg_a;
        bool valid_b = true;
        if (EPI == EPI_SWIGLU) {
            row_a = u; row_b = u; seg_a = 0;
            pa = w0 + (size_t)u * (size_t)n;
            pb = w1 + (size_t)u * (size_t)n;
            oa = out0; ob = out0;
        } else {
            const int ga = 2 * u;
            int gb = ga + 1;
            if (gb >= total_rows) { gb = ga; valid_b = false; }
            // segment select by comparison arithmetic on scalars (no indexed loads)
            const bool a1 = ga >= rows0, a2 = ga >= rows0 + rows1;
            const bool b1 = gb >= rows0, b2 = gb >= rows0 + rows1;
            row_a = ga - (a2 ? rows0 + rows1 : (a1 ? rows0 : 0));
            row_b = gb - (b2 ? rows0 + rows1 : (b1 ? rows0 : 0));
            seg_a = a2 ? 2 : (a1 ? 1 : 0);
            pa = (a2 ? w2 : (a1 ? w1 : w0)) + (size_t)row_a * (size_t)n;
            pb = (b2 ? w2 : (b1 ? w1 : w0)) + (size_t)row_b * (size_t)n;
            oa = a2 ? out2 + ps2 : (a1 ? out1 + ps1 : out0);
            ob = b2 ? out2 + ps2 : (b1 ? out1 + ps1 : out0);
        }
        float sa, sb;
        dot2<VEC>(pa, pb, xs, n, sa, sb);

        if (EPI == EPI_SWIGLU) {
            float v = sa;
            v = v * (1.0f / (1.0f + expf(-v)));  // :412
            v = v * sb;                          // :416
            if (lane == 0) oa[u] = v;
        } else if (EPI == EPI_ROPE) {
            // rows (row_a, row_a+1) of one segment: the pair (i, i+1) of :346-349
            float o0 = sa, o1 = sb;
            if (seg_a < a.rope_segs) {
                const int hs = a.head_size;
                const float2 cs = a.rope[(size_t)pos * (size_t)(hs >> 1) + (size_t)((row_a % hs) >> 1)];
                o0 = sa * cs.x - sb * cs.y;  // :348
                o1 = sa * cs.y + sb * cs.x;  // :349
            }
            if (lane == 0) {  // q, or the pos row of the K / V cache (:354-358)
                oa[row_a] = o0;
                if (valid_b) ob[row_b] = o1;
            }
        } else if (EPI == EPI_RESID) {
            if (lane == 0) {
                oa[row_a] = a.resid[row_a] + sa;  // :711 a[i] += b[i]
                if (valid_b) ob[row_b] = a.resid[row_b] + sb;
            }
        } else {
            if (lane == 0) {
                oa[row_a] = sa;
                if (valid_b) ob[row_b] = sb;
            }
        }
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

__host__ __device__ inline AttnGeom attn_geom(int head_size, bool vec)
{
    AttnGeom g;
    g.E = vec ? head_size >> 2 : head_size;
    int t = 1;
    while (t < g.E && t < 64) t <<= 1;
    g.TPR = t;
    g.G = kBlock / t;
    return g;
}

// scores for timesteps t < T: att[t] = dot(q, K[t]) / sqrt(head_size)   (:367-375)
template <bool VEC>
__device__ __forceinline__ void attn_scores(const float *qs, const float *__restrict__ kbase,
                                            int kv_stride, int head_size, int T, float div,
                                            float *att)
{
    const AttnGeom ge = attn_geom(head_size, VEC);
    const int g = threadIdx.x / ge.TPR, c0 = threadIdx.x % ge.TPR;
    for (int t = g; t < T; t += ge.G) {
        float p = 0.0f;
        const float *krow = kbase + (size_t)t * (size_t)kv_stride;
        if (VEC) {
            v4f acc = {0.f, 0.f, 0.f, 0.f};
            for (int c = c0; c < ge.E; c += ge.TPR)
                acc = fma4(((const v4f *)qs)[c], ((const v4f *)krow)[c], acc);
            p = hsum4(acc);
        } else {
            for (int c = c0; c < ge.E; c += ge.TPR) p = fmaf(qs[c], krow[c], p);
        }
        for (int o = ge.TPR >> 1; o > 0; o >>= 1) p += __shfl_xor(p, o, 64);
        if (c0 == 0) att[t] = p / div;  // :372 divide, not multiply by reciprocal
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
            for (int t = g; t < T; t += ge.G) {
                const v4f v = ((const v4f *)(vbase + (size_t)t * (size_t)kv_stride))[c];
                const float w = att[t];
                acc.x = fmaf(v.x, w, acc.x);
                acc.y = fmaf(v.y, w, acc.y);
                acc.z = fmaf(v.z, w, acc.z);
                acc.w = fmaf(v.w, w, acc.w);
            }
            ((v4f *)(part + (size_t)g * head_size))[c] = acc;
        } else {
            float acc = 0.0f;
            for (int t = g; t < T; t += ge.G)
                acc = fmaf(vbase[(size_t)t * (size_t)kv_stride + c], att[t], acc);
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
    for (int i = tid; i < a.vocab; i += blockDim.x) {
        const float v = a.logits[i];
        if (v > best || bi == 0x7fffffff) {  // strict '>' keeps the lowest index (:720)
            best = v;
            bi = i;
        }
    }
    // wave reduce: larger value wins, equal values -> lower index
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) {
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
            if (s_val[w] > best || (s_val[w] == best && s_idx[w] < bi)) {
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
template <bool VEC>
__global__ __launch_bounds__(kBlock) void rmsnorm_kernel(float *o, const float *x, const float *w,
                                                         int n)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *xs = lds, *scratch = lds + ((n + 3) & ~3);
    stage_x<PRO_RMS, VEC>(x, w, n, xs, scratch);
    for (int j = threadIdx.x; j < n; j += blockDim.x) o[j] = xs[j];
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

template <int PRO, int EPI>
hipError_t launch_matvec_pe(const MatvecArgs &a, bool vec, int grid, size_t lds, hipStream_t st)
{
    if (vec) {
        hipError_t e = ensure_lds(matvec_kernel<PRO, EPI, true>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((matvec_kernel<PRO, EPI, true>), dim3(grid), dim3(kBlock), lds, st, a);
    } else {
        hipError_t e = ensure_lds(matvec_kernel<PRO, EPI, false>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((matvec_kernel<PRO, EPI, false>), dim3(grid), dim3(kBlock), lds, st, a);
    }
    return hipGetLastError();
}

}  // namespace

size_t matvec_lds_bytes(int n) { return (size_t)(((n + 3) & ~3) + kScratch) * sizeof(float); }

size_t attention_lds_bytes(int head_size, int seq_len, bool vec)
{
    const AttnGeom ge = attn_geom(head_size, vec);
    return (size_t)(((head_size + 3) & ~3) + ((seq_len + 3) & ~3) + ge.G * head_size + kScratch) *
           sizeof(float);
}

hipError_t launch_matvec(const MatvecArgs &a, int pro, int epi, int max_blocks, hipStream_t st)
{
    bool vec = (a.n % 4) == 0 && aligned16(a.x) && aligned16(a.w0);
    if (a.rows1 > 0) vec = vec && aligned16(a.w1);
    if (a.rows2 > 0) vec = vec && aligned16(a.w2);
    if (pro == PRO_RMS) vec = vec && aligned16(a.rms_w);
    if (epi == EPI_SWIGLU && a.rows1 != a.rows0) return hipErrorInvalidValue;
    const int total_rows = a.rows0 + a.rows1 + a.rows2;
    const int n_units = (epi == EPI_SWIGLU) ? a.rows0 : (total_rows + 1) / 2;
    if (n_units <= 0) return hipErrorInvalidValue;
    // Even split: every wave gets the same number of units (+-1).
    int blocks_needed = (n_units + kWaves - 1) / kWaves;
    int grid = blocks_needed;
    if (grid > max_blocks) {
        const int per_wave = (n_units + max_blocks * kWaves - 1) / (max_blocks * kWaves);
        grid = (n_units + per_wave * kWaves - 1) / (per_wave * kWaves);
    }
    const size_t lds = matvec_lds_bytes(a.n);
#define L2Z_MV(P, E) \
    if (pro == P && epi == E) return launch_matvec_pe<P, E>(a, vec, grid, lds, st);
    L2Z_MV(PRO_NONE, EPI_STORE)
    L2Z_MV(PRO_NONE, EPI_RESID)
    L2Z_MV(PRO_RMS, EPI_STORE)
    L2Z_MV(PRO_RMS, EPI_ROPE)
    L2Z_MV(PRO_RMS, EPI_SWIGLU)
#undef L2Z_MV
    return hipErrorInvalidValue;
}

hipError_t launch_attention(const AttnArgs &a, int n_heads_local, hipStream_t st)
{
    const bool vec = (a.head_size % 4) == 0 && (a.kv_dim % 4) == 0 && aligned16(a.q) &&
                     aligned16(a.kcache) && aligned16(a.vcache);
    const size_t lds = attention_lds_bytes(a.head_size, a.seq_len, vec);
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
    hipFree(part);
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

}  // namespace l2z
