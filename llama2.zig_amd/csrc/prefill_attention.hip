// prefill_attention.hip -- the non-GEMM kernels of the batched prompt pass: batched rmsnorm, embedding
// gather, causal attention of a chunk (one block per (head, token) with the decode kernel's arithmetic,
// or one block per (head, 64 queries) on the fp32 matrix cores).
#include "prefill_common.h"

namespace l2z {
namespace {

// rows of x -> rmsnorm rows (main.zig:432-468), one block per token
__global__ __launch_bounds__(kPfBlock) void prefill_rmsnorm(float *o, const float *x, const float *w,
                                                            int n, int P)
{
    __shared__ float red[8];
    const int t = blockIdx.x;
    const float *xr = x + (size_t)t * n;
    const int n4 = n >> 2;  // n % 4 == 0 on this path
    constexpr int R = 8;    // float4 kept in registers per lane: one round trip up to n = 8192
    const bool in_regs = n4 <= R * kPfBlock;
    v4f xv[R];
    float ss = 0.0f;
    if (in_regs) {
#pragma unroll
        for (int k = 0; k < R; k++) {
            const int i = threadIdx.x + kPfBlock * k;
            xv[k] = i < n4 ? ((const v4f *)xr)[i] : v4f{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < R; k++) {
            ss = fmaf(xv[k].x, xv[k].x, ss); ss = fmaf(xv[k].y, xv[k].y, ss);
            ss = fmaf(xv[k].z, xv[k].z, ss); ss = fmaf(xv[k].w, xv[k].w, ss);
        }
    } else {
        for (int i = threadIdx.x; i < n; i += blockDim.x) ss = fmaf(xr[i], xr[i], ss);
    }
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    float tot = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); i++) tot += red[i];
    float s = tot / (float)n;  // main.zig:452-455
    s += 1e-5f;
    s = 1.0f / sqrtf(s);
    if (in_regs) {
#pragma unroll
        for (int k = 0; k < R; k++) {
            const int i = threadIdx.x + kPfBlock * k;
            if (i < n4) {
                const v4f wv = ((const v4f *)w)[i];
                v4f r;
                r.x = (xv[k].x * s) * wv.x; r.y = (xv[k].y * s) * wv.y;
                r.z = (xv[k].z * s) * wv.z; r.w = (xv[k].w * s) * wv.w;
                ((v4f *)(o + (size_t)t * n))[i] = r;
            }
        }
    } else {
        for (int i = threadIdx.x; i < n; i += blockDim.x) o[(size_t)t * n + i] = (xr[i] * s) * w[i];
    }
}

// x[t] = embedding row of tokens[t]   (main.zig:295-296)
__global__ void prefill_embed(float *x, const float *tok_emb, const int *tokens, int dim)
{
    const float *row = tok_emb + (size_t)tokens[blockIdx.x] * dim;
    for (int i = threadIdx.x; i < dim; i += blockDim.x) x[(size_t)blockIdx.x * dim + i] = row[i];
}

// Causal attention for a chunk (main.zig:361-389): block (h, t) is query token t of head h and
// attends to cache rows 0..pos0+t.  256 threads = G groups of TPR lanes, as in the decode kernel.
__global__ __launch_bounds__(kPfBlock) void prefill_attention(const float *q, int ldq,
                                                              const float *kcache, const float *vcache,
                                                              float *out, int ldo, int pos0,
                                                              int head_size, int kv_dim, int kv_mul,
                                                              int seq_len)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int hs = head_size, E = hs >> 2;
    int TPR = 1;
    while (TPR < E && TPR < 64) TPR <<= 1;
    const int G = kPfBlock / TPR;
    float *att = lds;                                   // seq_len
    float *part = att + ((seq_len + 3) & ~3);           // G*hs
    float *red = part + (size_t)G * hs;                 // 8
    const int h = blockIdx.x, tok = blockIdx.y;
    const int T = pos0 + tok + 1;
    const int kvh = h / kv_mul;
    const float *kbase = kcache + (size_t)kvh * hs, *vbase = vcache + (size_t)kvh * hs;
    const int g = threadIdx.x / TPR, c0 = threadIdx.x % TPR;
    const bool active = c0 < E;
    const int cc = active ? c0 : 0;
    const v4f zero = {0.f, 0.f, 0.f, 0.f};
    const v4f qv = active ? ((const v4f *)(q + (size_t)tok * ldq + (size_t)h * hs))[cc] : zero;
    const float div = sqrtf((float)hs);
    for (int t = g; t < T; t += G) {
        const v4f kv = ((const v4f *)(kbase + (size_t)t * kv_dim))[cc];
        float p = fmaf(qv.x, kv.x, 0.0f);
        p = fmaf(qv.y, kv.y, p); p = fmaf(qv.z, kv.z, p); p = fmaf(qv.w, kv.w, p);
        for (int o = TPR >> 1; o > 0; o >>= 1) p += __shfl_xor(p, o, 64);
        if (c0 == 0) att[t] = p / div;
    }
    __syncthreads();
    float m = -INFINITY;
    for (int t = threadIdx.x; t < T; t += blockDim.x) m = fmaxf(m, att[t]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.0f;
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        const float e = expf(att[t] - m);
        att[t] = e;
        s += e;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = s;
    __syncthreads();
    s = ((red[4] + red[5]) + red[6]) + red[7];
    v4f acc = zero;
    for (int t = g; t < T; t += G) {
        const v4f vv = ((const v4f *)(vbase + (size_t)t * kv_dim))[cc];
        const float w = att[t] / s;  // main.zig:704
        acc.x = fmaf(vv.x, w, acc.x); acc.y = fmaf(vv.y, w, acc.y);
        acc.z = fmaf(vv.z, w, acc.z); acc.w = fmaf(vv.w, w, acc.w);
    }
    if (active) ((v4f *)(part + (size_t)g * hs))[cc] = acc;
    __syncthreads();
    for (int i = threadIdx.x; i < hs; i += blockDim.x) {
        float r = part[i];
        for (int gg = 1; gg < G; gg++) r += part[(size_t)gg * hs + i];
        out[(size_t)tok * ldo + (size_t)h * hs + i] = r;
    }
}

// Tiled causal attention for a chunk (flash form): block (h, 64 query tokens) walks the cache in
// tiles of 64 timesteps; S = Q K^T and O += P V run on the fp32 matrix cores, the softmax is the
// running-max form (m, l per query row, O rescaled by e^(m_old - m_new)), so a K/V tile is read
// once per 64 queries instead of once per query.  Mathematically main.zig:361-389; in floating
// point the weights are e^(s-m)/l applied after the sum instead of before (a few ulp, same as the
// decode path's split attention).  LDS: Q, K, V tiles 64 x (hs+1), P tile 64 x 65.
//   S: wave (wm, wn) owns S[32 wm.., 32 wn..];  O: 32 x 32 tiles (row half, column tile) dealt to
//   the waves round-robin, TPW per wave.
template <int TPW>
__global__ __launch_bounds__(kPfBlock) void prefill_attention_tiled(const float *q, int ldq,
                                                                    const float *kcache, const float *vcache,
                                                                    float *out, int ldo, int pos0, int P,
                                                                    int hs, int kv_dim, int kv_mul, int seq_len)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int LD = hs + 1;
    float *qs = lds, *ks = qs + 64 * LD, *vs = ks + 64 * LD, *ps = vs + 64 * LD;
    float *row_m = ps + 64 * 65, *row_l = row_m + 64, *row_a = row_l + 64;
    const int h = blockIdx.x, q0 = blockIdx.y * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int kvh = h / kv_mul;  // :369
    const float *kbase = kcache + (size_t)kvh * hs, *vbase = vcache + (size_t)kvh * hs;
    const int E = hs >> 2;
    const v4f zero = {0.f, 0.f, 0.f, 0.f};
    for (int f = tid; f < 64 * E; f += kPfBlock) {
        const int r = f / E, c = (f % E) * 4;
        const v4f v = q0 + r < P ? *(const v4f *)(q + (size_t)(q0 + r) * ldq + (size_t)h * hs + c) : zero;
        float *d = qs + r * LD + c;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    if (tid < 64) { row_m[tid] = -INFINITY; row_l[tid] = 0.0f; }
    v16f acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][r] = 0.0f;
    const int last_q = (q0 + 63 < P ? q0 + 63 : P - 1);
    const int n_kt = (pos0 + last_q) / 64 + 1;  // key tiles 0 .. the one holding the last query's own position
    const float div = sqrtf((float)hs);
    // k order of the S product: pairs 32 apart inside chunks of 64 when hs allows (lanes 32..63 then
    // hit LDS banks 32 away from lanes 0..31), else pairs hs/2 apart
    const bool chunked = (hs & 63) == 0;
    const int half = hs >> 1;
    for (int kt = 0; kt < n_kt; kt++) {
        const int t0 = kt * 64;
        __syncthreads();  // the previous tile's P V product is done with ks / vs / ps
        for (int f = tid; f < 64 * E; f += kPfBlock) {
            const int r = f / E, c = (f % E) * 4;
            int t = t0 + r;
            t = t < seq_len ? t : seq_len - 1;  // rows past the context are masked below
            const v4f kv = *(const v4f *)(kbase + (size_t)t * kv_dim + c);
            const v4f vv = *(const v4f *)(vbase + (size_t)t * kv_dim + c);
            float *dk = ks + r * LD + c, *dv = vs + r * LD + c;
            dk[0] = kv.x; dk[1] = kv.y; dk[2] = kv.z; dk[3] = kv.w;
            dv[0] = vv.x; dv[1] = vv.y; dv[2] = vv.z; dv[3] = vv.w;
        }
        __syncthreads();
        {   // S = Q K^T for this wave's 32 x 32 tile (:367-371)
            v16f sacc;
#pragma unroll
            for (int r = 0; r < 16; r++) sacc[r] = 0.0f;
            const float *qa = qs + (32 * wm + li) * LD, *kb = ks + (32 * wn + li) * LD;
            for (int st = 0; st < half; st++) {
                const int k = chunked ? ((st >> 5) << 6) + (st & 31) + 32 * lh : st + half * lh;
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[k], kb[k], sacc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * lh, col = 32 * wn + li;
                const bool live = t0 + col <= pos0 + q0 + row;  // causal: t <= pos of the query
                ps[row * 65 + col] = live ? sacc[r] / div : -INFINITY;  // :372
            }
        }
        __syncthreads();
        {   // running softmax, 4 lanes per query row (:687-706 in running-max form)
            const int row = tid >> 2, q4 = tid & 3;
            float *pr = ps + row * 65;
            float mx = -INFINITY;
            for (int c = q4; c < 64; c += 4) mx = fmaxf(mx, pr[c]);
            mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
            const float m_old = row_m[row];
            const float m_new = fmaxf(m_old, mx);  // finite: key 0 is live for every query
            float sum = 0.0f;
            for (int c = q4; c < 64; c += 4) {
                const float e = expf(pr[c] - m_new);
                pr[c] = e;
                sum += e;
            }
            sum += __shfl_xor(sum, 1, 64);
            sum += __shfl_xor(sum, 2, 64);
            if (q4 == 0) {
                const float alpha = expf(m_old - m_new);
                row_a[row] = alpha;
                row_l[row] = row_l[row] * alpha + sum;
                row_m[row] = m_new;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TPW; j++) {  // O = O e^(m_old - m_new) + P V  (:381-388)
            const int ti = wave + 4 * j, rt = ti & 1, ct = ti >> 1;
            if (ct * 32 < hs) {
#pragma unroll
                for (int r = 0; r < 16; r++) acc[j][r] *= row_a[32 * rt + (r & 3) + 8 * (r >> 2) + 4 * lh];
                const float *pa = ps + (32 * rt + li) * 65 + 32 * lh;
                const float *vb = vs + (32 * lh) * LD + 32 * ct + li;  // columns >= hs: finite junk, dropped
#pragma unroll 8
                for (int st = 0; st < 32; st++)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[st], vb[st * LD], acc[j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < TPW; j++) {
        const int ti = wave + 4 * j, rt = ti & 1, ct = ti >> 1;
        const int col = 32 * ct + li;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (col < hs && q0 + row < P)
                out[(size_t)(q0 + row) * ldo + (size_t)h * hs + col] = acc[j][r] / row_l[row];  // :704
        }
    }
}

}  // namespace

hipError_t launch_prefill_rmsnorm(float *o, const float *x, const float *w, int n, int P,
                                  hipStream_t st)
{
    hipLaunchKernelGGL(prefill_rmsnorm, dim3(P), dim3(kPfBlock), 0, st, o, x, w, n, P);
    return hipGetLastError();
}

hipError_t launch_prefill_embed(float *x, const float *tok_emb, const int *tokens, int dim, int P,
                                hipStream_t st)
{
    hipLaunchKernelGGL(prefill_embed, dim3(P), dim3(256), 0, st, x, tok_emb, tokens, dim);
    return hipGetLastError();
}

hipError_t launch_prefill_attention(const float *q, int ldq, const float *kcache, const float *vcache,
                                    float *out, int ldo, int pos0, int P, int n_heads, int head_size,
                                    int kv_dim, int kv_mul, int seq_len, hipStream_t st, int n_heads_model)
{
    // the two kernels round differently; a shard must take the one the unsharded pass takes
    if (n_heads_model <= 0) n_heads_model = n_heads;
    const bool naive = tunables().pf_attn == 0;
    const size_t lds_t = (size_t)(3 * 64 * (head_size + 1) + 64 * 65 + 3 * 64) * sizeof(float);
    const int n_ct = (head_size + 31) / 32;  // O column tiles; 2 n_ct tiles over 4 waves
    // one block per (head, 64 queries): worth it once that fills half the CUs (7B: from 256 tokens);
    // below, and for models with few heads, the block-per-(head, query) kernel has more parallelism
    const bool enough_blocks = n_heads_model * ((P + 63) / 64) >= 128;
    if (!naive && enough_blocks && lds_t <= 160 * 1024 && n_ct <= 8 && (head_size % 4) == 0 && (kv_dim % 4) == 0) {
        const int tpw = (2 * n_ct + 3) / 4;
        const void *fn = tpw <= 1 ? (const void *)prefill_attention_tiled<1>
                       : tpw == 2 ? (const void *)prefill_attention_tiled<2>
                                  : (const void *)prefill_attention_tiled<4>;
        if (lds_t > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t);
            if (e != hipSuccess) return e;
        }
        const dim3 grid(n_heads, (P + 63) / 64);
        if (tpw <= 1)
            hipLaunchKernelGGL(prefill_attention_tiled<1>, grid, dim3(kPfBlock), lds_t, st, q, ldq, kcache, vcache,
                               out, ldo, pos0, P, head_size, kv_dim, kv_mul, seq_len);
        else if (tpw == 2)
            hipLaunchKernelGGL(prefill_attention_tiled<2>, grid, dim3(kPfBlock), lds_t, st, q, ldq, kcache, vcache,
                               out, ldo, pos0, P, head_size, kv_dim, kv_mul, seq_len);
        else
            hipLaunchKernelGGL(prefill_attention_tiled<4>, grid, dim3(kPfBlock), lds_t, st, q, ldq, kcache, vcache,
                               out, ldo, pos0, P, head_size, kv_dim, kv_mul, seq_len);
        return hipGetLastError();
    }
    int E = head_size >> 2, TPR = 1;
    while (TPR < E && TPR < 64) TPR <<= 1;
    const int G = kPfBlock / TPR;
    const size_t lds = (size_t)(((seq_len + 3) & ~3) + G * head_size + 8) * sizeof(float);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(prefill_attention),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(prefill_attention, dim3(n_heads, P), dim3(kPfBlock), lds, st, q, ldq, kcache,
                       vcache, out, ldo, pos0, head_size, kv_dim, kv_mul, seq_len);
    return hipGetLastError();
}

}  // namespace l2z
