// prefill_attention.hip -- the non-GEMM kernels of the batched prompt pass: batched rmsnorm, embedding
// gather, causal attention of a chunk (one block per (head, token) with the decode kernel's arithmetic,
// or one block per (head, 64 queries) on the fp32 matrix cores).
#include <type_traits>

#include "prefill_common.h"

namespace l2z {
namespace {

// rows of x -> rmsnorm rows (main.zig:432-468), one block per token
// x3 != null: the rows' planes of bf16 terms too (x3[token][plane][kp]: what the GEMM on the bf16 matrix cores reads;
// prefill_common.h) -- the consumer's own split launch is then not needed
// pend.valid: the residual product before this norm (Wo / W2 on the stream form) left its K ranges' sums in the workspace:
// x[t][f] += range 0 + range 1 + ... (in that order, then the residual: the sums its own hand-over forms) first, and x is
// written back (l2z_internal.h DeferredSum; the sums lie as the blocks' MFMA accumulators held them: prefill_gemm.hip)
__device__ __forceinline__ const float *deferred_at(const DeferredSum &d, int t, int f)
{
    const int bx = f / d.feat, fw = f - bx * d.feat, row = t & 31;
    const int lane = (fw & 31) + 32 * ((row >> 2) & 1), r = (row & 3) + 4 * (row >> 3);
    return d.part + ((size_t)bx * d.sk * (d.feat >> 5) + (fw >> 5)) * (size_t)(d.tm * 16 * 64) + (size_t)(((t >> 5) * 16 + r) * 64 + lane);
}

__global__ __launch_bounds__(kPfBlock) void prefill_rmsnorm(float *o, int ldo, float *x, const float *w,
                                                            int n, int P, __bf16 *x3, int kp, const DeferredSum pend)
{
    __shared__ float red[8];
    const int t = blockIdx.x;
    float *xr = x + (size_t)t * n;
    const int n4 = n >> 2;  // n % 4 == 0 on this path
    constexpr int R = 8;    // float4 kept in registers per lane: one round trip up to n = 8192
    const bool in_regs = n4 <= R * kPfBlock;
    const size_t zstride = (size_t)(pend.feat >> 5) * (size_t)(pend.tm * 16 * 64);   // floats between a tile's ranges
    v4f xv[R];
    float ss = 0.0f;
    if (in_regs) {
#pragma unroll
        for (int k = 0; k < R; k++) {
            const int i = threadIdx.x + kPfBlock * k;
            xv[k] = i < n4 ? ((const v4f *)xr)[i] : v4f{0.f, 0.f, 0.f, 0.f};
        }
        if (pend.valid) {
            // (the ranges of a float4 requested together -- a literal count per case -- then added in order)
            auto add_ranges = [&](auto sk_c) {
                constexpr int SK = decltype(sk_c)::value;
#pragma unroll
                for (int k = 0; k < R; k++) {
                    const int i = threadIdx.x + kPfBlock * k;
                    if (i < n4) {
                        const float *p = deferred_at(pend, t, 4 * i);   // four features = four adjacent lanes of one wave's fragment
                        v4f pv[SK];
#pragma unroll
                        for (int z = 0; z < SK; z++) pv[z] = *(const v4f *)(p + z * zstride);
                        v4f v = pv[0];
#pragma unroll
                        for (int z = 1; z < SK; z++) v += pv[z];
                        xv[k] = xv[k] + v;   // main.zig:711
                        ((v4f *)xr)[i] = xv[k];
                    }
                }
            };
            if (pend.sk == 2) add_ranges(std::integral_constant<int, 2>{});
            else if (pend.sk == 4) add_ranges(std::integral_constant<int, 4>{});
            else add_ranges(std::integral_constant<int, 8>{});
        }
#pragma unroll
        for (int k = 0; k < R; k++) {
            ss = fmaf(xv[k].x, xv[k].x, ss); ss = fmaf(xv[k].y, xv[k].y, ss);
            ss = fmaf(xv[k].z, xv[k].z, ss); ss = fmaf(xv[k].w, xv[k].w, ss);
        }
    } else {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            if (pend.valid) {
                const float *p = deferred_at(pend, t, i);
                float v = p[0];
                for (int z = 1; z < pend.sk; z++) v += p[z * zstride];
                xr[i] = xr[i] + v;
            }
            ss = fmaf(xr[i], xr[i], ss);
        }
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
                ((v4f *)(o + (size_t)t * ldo))[i] = r;
                if (x3) planes_store4(x3, kp, t, 4 * i, r);
            }
        }
    } else {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const float r = (xr[i] * s) * w[i];
            o[(size_t)t * ldo + i] = r;
            if (x3) planes_store1(x3, kp, t, i, r);
        }
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
                                                              int head_size, int kv_dim, size_t kv_head, int kv_mul,
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
    const float *kbase = kcache + (size_t)kvh * kv_head, *vbase = vcache + (size_t)kvh * kv_head;  // kv_dim: row stride
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
// Causal attention of a chunk, one block per (head, 64 queries), flash form (round 2): each of the four
// waves owns 16 queries for the whole kernel -- their running max / sum and the O accumulators never
// leave its registers, and no block-wide step exists except bringing in the next 64 key / value rows.
// Everything is computed TRANSPOSED on MFMA 16x16x4 f32 (D[i][j] += A[i][k] B[k][j]; A: lane l holds
// A[l & 15][l >> 4], B: lane l holds B[l >> 4][l & 15], D: lane l, register r holds D[4 (l >> 4) + r][l & 15]):
//   S^T[kv][q]  = sum_h K[kv][h] Q[q][h]        A = K rows (LDS), B = Q (registers, loaded once)
//   O^T[d][q]  += sum_kv V[kv][d] P^T[kv][q]    A = V (LDS),      B = P^T
// In the D layout a lane holds S^T for ONE query (q = l & 15) and four key rows (4 (l >> 4) + r) per tile;
// exactly the values P^T's B operand wants from that lane if the k index of the PV product runs over
// kv' = 4 k + s instead of 4 s + k -- a sum over kv has no order in exact arithmetic and V's operand uses
// the same kv' -- so P goes from the S accumulators into the PV product without leaving the registers:
// no transpose, no LDS round trip, no barrier.  A query's softmax statistics: the lane's 16 values, then
// two shuffles across the four lane groups.  The per-query rescale of O^T is a per-lane scalar.
// K and V tiles come in by direct-to-LDS loads.  Operands are read as ds_read_b128, one float4 feeding
// four MFMAs: K's float4 (4 consecutive h) with Q's float4 of the same h -- k-set {16 T + 4 g + c};
// V's float4 (4 consecutive d) feeds four output tiles, so O^T tile (DT, c) row i is d = 64 DT + 4 i + c.
// K rows are read 16 rows x one slot at a time: physical slot = logical ^ (row & 15), applied on the
// source address of the load; V rows are read a row at a time (16 consecutive slots): no swizzle.
// head_size = 16 NDT, NDT in {4, 8}; two buffers of 64 KB at head_size 128: the next tile travels while
// this one is multiplied, one barrier per tile.
// Scores are x / sqrt(head_size) as in main.zig:372; masked scores are -inf: exp gives exactly 0.
template <int NDT, int KH>
__global__ __launch_bounds__(256 * KH) void prefill_attention_flash(const float *q, int ldq, const float *kcache,
                                                                    const float *vcache, float *out, int ldo,
                                                                    int pos0, int P, int kv_dim, size_t kv_head, int kv_mul,
                                                                    int seq_len, __bf16 *x3, int kp)
{
    constexpr int HS = 16 * NDT, E = HS / 4;  // float4 slots per row
    constexpr int NWV = 4 * KH;               // waves: 4 query groups x KH parts of every key tile
    constexpr int JT = 4 / KH;                // 16-row key sub-tiles per wave and tile
    constexpr int LPW = 4 * NDT / NWV;        // wave-wide loads per wave, tile and matrix (64 E slots / 64 / NWV)
    static_assert((NDT == 4 || NDT == 8) && (KH == 1 || KH == 2), "head_size 64 or 128; one or two key parts");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // blocks are dispatched in id order: the query tiles with the most key tiles (the LAST queries) first
    const int h = blockIdx.x, q0 = ((int)gridDim.y - 1 - (int)blockIdx.y) * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qg = wave & 3, kh = wave >> 2;  // query group (16 queries), key part (rows 64 / KH * kh .. of a tile)
    const int qi = lane & 15, g = lane >> 4;
    const int kvh = h / kv_mul;  // :369
    const float *kbase = kcache + (size_t)kvh * kv_head, *vbase = vcache + (size_t)kvh * kv_head;  // kv_dim: row stride
    const int myq = q0 + 16 * qg + qi;              // this lane's query (token index in the chunk)
    const int qrow = myq < P ? myq : P - 1;         // past the chunk: a valid row, results dropped
    v4f qreg[NDT];
#pragma unroll
    for (int T = 0; T < NDT; T++) qreg[T] = *(const v4f *)(q + (size_t)qrow * ldq + (size_t)h * HS + 16 * T + 4 * g);
    const v4f zero = {0.f, 0.f, 0.f, 0.f};
    v4f ot[NDT];
#pragma unroll
    for (int d = 0; d < NDT; d++) ot[d] = zero;
    float m = -INFINITY, lsum = 0.0f;
    const int last_q = (q0 + 63 < P ? q0 + 63 : P - 1);
    const int n_kt = (pos0 + last_q) / 64 + 1;              // key tiles of the block
    const int last_live = pos0 + q0 + 16 * qg + 15;         // last key position live for one of this wave's queries
    const float div = sqrtf((float)HS);
    // two K / V buffers: tile kt + 1 travels while tile kt is multiplied; one barrier per tile
    auto issue = [&](int kt, int buf) {
        const int t0 = kt * 64;
        float *kd = lds + buf * (2 * 64 * HS), *vd = kd + 64 * HS;
#pragma unroll
        for (int i = 0; i < LPW; i++) {  // 64 E float4 slots per matrix, 64 per wave-wide load
            const int f = (wave * LPW + i) * 64 + lane, row = f / E, cp = f % E;
            int t = t0 + row;
            t = t < seq_len ? t : seq_len - 1;  // rows past the context are masked below
            lds_dma16(kbase + (size_t)t * kv_dim + 4 * (cp ^ (row & 15)), kd + (wave * LPW + i) * 256);
            lds_dma16(vbase + (size_t)t * kv_dim + 4 * cp, vd + (wave * LPW + i) * 256);
        }
    };
    issue(0, 0);
    for (int kt = 0; kt < n_kt; kt++) {
        const int t0 = kt * 64;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of tile kt has landed
        __syncthreads();  // everyone's has; and every wave is done with tile kt - 1: its buffer is free
        if (kt + 1 < n_kt) issue(kt + 1, (kt + 1) & 1);
        const float *ks = lds + (kt & 1) * (2 * 64 * HS), *vs = ks + 64 * HS;
        const int r0 = (64 / KH) * kh;                 // this wave's rows of the tile: r0 .. r0 + 16 JT - 1
        if (t0 + r0 > last_live) continue;             // nothing live for this wave (wave-uniform)
        v4f st[JT];
#pragma unroll
        for (int jt = 0; jt < JT; jt++) st[jt] = zero;
#pragma unroll
        for (int T = 0; T < NDT; T++)  // the JT accumulators interleaved: independent MFMA chains
#pragma unroll
            for (int jt = 0; jt < JT; jt++) {
                const int row = r0 + 16 * jt + qi;
                const v4f kq = ((const v4f *)(ks + row * HS))[(4 * T + g) ^ (row & 15)];
#pragma unroll
                for (int c = 0; c < 4; c++)
                    st[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kq[c], qreg[T][c], st[jt], 0, 0, 0);
            }
        // st[jt][r] = S^T[key t0 + r0 + 16 jt + 4 g + r][query myq]: scale, causal mask, running softmax
        float mx = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < JT; jt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const bool live = t0 + r0 + 16 * jt + 4 * g + r <= pos0 + myq;  // t <= pos of the query
                st[jt][r] = live ? st[jt][r] / div : -INFINITY;                 // :372
                mx = fmaxf(mx, st[jt][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m, mx);
        // a query none of whose keys this wave has seen live yet (KH = 2: the upper key part of the first
        // tiles): every weight is e^(-inf) = 0 against any finite reference
        const float mref = m_new == -INFINITY ? 0.0f : m_new;
        float sum = 0.0f;
#pragma unroll
        for (int jt = 0; jt < JT; jt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                st[jt][r] = expf(st[jt][r] - mref);
                sum += st[jt][r];
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float alpha = expf(m - mref);  // first live tile: e^(-inf) = 0
        lsum = lsum * alpha + sum;
        m = m_new;
#pragma unroll
        for (int d = 0; d < NDT; d++) ot[d] *= alpha;
        // O^T += V^T P^T, k index of step (jt, s) = key rows r0 + 16 jt + 4 k + s (:381-388)
#pragma unroll
        for (int DT = 0; DT < NDT / 4; DT++)
#pragma unroll
            for (int jt = 0; jt < JT; jt++)
#pragma unroll
                for (int s4 = 0; s4 < 4; s4++) {
                    const v4f vq = *(const v4f *)(vs + (r0 + 16 * jt + 4 * g + s4) * HS + 4 * (16 * DT + qi));
#pragma unroll
                    for (int c = 0; c < 4; c++)
                        ot[4 * DT + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(vq[c], st[jt][s4], ot[4 * DT + c], 0, 0, 0);
                }
    }
    if (KH == 2) {
        // the two key parts of a query group: merge (m, l, O) of the upper part into the lower one
        // (main.zig:687-706 is one softmax over all keys: rescale both parts to the common maximum)
        __syncthreads();  // K / V buffers are free
        float *mg = lds + (size_t)(qg * 64 + lane) * (4 * NDT + 2);
        if (kh == 1) {
#pragma unroll
            for (int d = 0; d < NDT; d++)
#pragma unroll
                for (int r = 0; r < 4; r++) mg[4 * d + r] = ot[d][r];
            mg[4 * NDT] = m;
            mg[4 * NDT + 1] = lsum;
        }
        __syncthreads();
        if (kh == 1) return;
        const float m1 = mg[4 * NDT], l1 = mg[4 * NDT + 1];
        const float mm = fmaxf(m, m1);  // finite: the lower part holds key 0
        const float a0 = expf(m - mm), a1 = expf(m1 - mm);
        lsum = lsum * a0 + l1 * a1;
#pragma unroll
        for (int d = 0; d < NDT; d++)
#pragma unroll
            for (int r = 0; r < 4; r++) ot[d][r] = ot[d][r] * a0 + mg[4 * d + r] * a1;
    }
    // ot[4 DT + c][r] = O^T[d = 64 DT + 16 g + 4 r + c][query myq]: 16 consecutive d per (lane, DT)
    if (myq < P) {
        float *o = out + (size_t)myq * ldo + (size_t)h * HS;
#pragma unroll
        for (int DT = 0; DT < NDT / 4; DT++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                v4f v;
#pragma unroll
                for (int c = 0; c < 4; c++) v[c] = ot[4 * DT + c][r] / lsum;  // :704
                *(v4f *)(o + 64 * DT + 16 * g + 4 * r) = v;
                if (x3) planes_store4(x3, kp, myq, h * HS + 64 * DT + 16 * g + 4 * r, v);   // (the Wo product's planes: see prefill_rmsnorm)
            }
    }
}

template <int TPW>
__global__ __launch_bounds__(kPfBlock) void prefill_attention_tiled(const float *q, int ldq,
                                                                    const float *kcache, const float *vcache,
                                                                    float *out, int ldo, int pos0, int P,
                                                                    int hs, int kv_dim, size_t kv_head, int kv_mul, int seq_len)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int LD = hs + 1;
    float *qs = lds, *ks = qs + 64 * LD, *vs = ks + 64 * LD, *ps = vs + 64 * LD;
    float *row_m = ps + 64 * 65, *row_l = row_m + 64, *row_a = row_l + 64;
    // blocks are dispatched in id order: the query tiles with the most key tiles (the LAST queries: causal)
    // go first, so that a grid of several waves does not end on its heaviest blocks
    const int h = blockIdx.x, q0 = ((int)gridDim.y - 1 - (int)blockIdx.y) * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int kvh = h / kv_mul;  // :369
    const float *kbase = kcache + (size_t)kvh * kv_head, *vbase = vcache + (size_t)kvh * kv_head;  // kv_dim: row stride
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

hipError_t launch_prefill_rmsnorm(float *o, int ldo, const float *x, const float *w, int n, int P,
                                  hipStream_t st, void *x3, int kp, const DeferredSum *pending)
{
    if (x3 != nullptr && (kp != n || (n & 3))) return hipErrorInvalidValue;   // (pad columns: the split launch writes their zeros)
    DeferredSum pend = {nullptr, 1, 128, 1, false};
    if (pending != nullptr && pending->valid) {
        if (pending->part == nullptr || (pending->sk != 2 && pending->sk != 4 && pending->sk != 8) || (pending->feat & 31) || pending->tm * 32 < P)
            return hipErrorInvalidValue;
        pend = *pending;
    }
    hipLaunchKernelGGL(prefill_rmsnorm, dim3(P), dim3(kPfBlock), 0, st, o, ldo, const_cast<float *>(x), w, n, P, (__bf16 *)x3, kp, pend);
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
                                    int kv_dim, size_t kv_head, int kv_mul, int seq_len, hipStream_t st, int n_heads_model,
                                    int form, void *x3, int kp, bool *planes_written)
{
    if (planes_written) *planes_written = false;
    // kv_dim here: floats between consecutive timesteps of one kv head (head-major cache: head_size)
    // the kernels round differently; a shard must take the one the unsharded pass takes
    // form (the test hook l2z_prefill_attention): 0 by shape, 1 block per (head, query), 2 tiled, 3 flash
    if (n_heads_model <= 0) n_heads_model = n_heads;
    const bool naive = form == 1;
    const size_t lds_t = (size_t)(3 * 64 * (head_size + 1) + 64 * 65 + 3 * 64) * sizeof(float);
    const int n_ct = (head_size + 31) / 32;  // O column tiles; 2 n_ct tiles over 4 waves
    // one block per (head, 64 queries): worth it once that fills half the CUs (7B: from 256 tokens);
    // below, and for models with few heads, the block-per-(head, query) kernel has more parallelism
    const bool enough_blocks = form == 2 || n_heads_model * ((P + 63) / 64) >= 128;
    // the flash form is ahead of the block-per-(head, query) kernel from far fewer blocks (12 heads x 4 query
    // tiles, stories110M at 256 tokens: 22 -> 16 us)
    // ... but not for chunks of <= 32 tokens: 32 heads x one query tile of mostly masked rows is a 12.5 us latency
    // chain, the per-query blocks (heads x tokens of them) take ~8 (7B shape, 16-token prompt: 5.74 -> 5.60 ms,
    // 4 tokens 5.89 -> 5.73; profiles/r03_prefill_short_ab.txt)
    const bool flash_blocks = form == 3 || (n_heads_model * ((P + 63) / 64) >= 32 && P > 32);
    if (!naive && form != 2 && flash_blocks && (head_size == 64 || head_size == 128) && (kv_dim % 4) == 0 &&
        (ldq % 4) == 0 && (ldo % 4) == 0 && (((uintptr_t)q | (uintptr_t)out | (uintptr_t)kcache | (uintptr_t)vcache) & 15) == 0) {
        // flash form: head sizes 64 and 128 (other head sizes: the LDS-softmax tiled kernel below)
        const size_t lds_f = (size_t)2 * 2 * 64 * head_size * sizeof(float);  // two buffers of a K and a V tile
        // two key parts per tile (8 waves: two per SIMD cover each other's latencies; one part, 4 waves, measured
        // slower at every length and removed in round 6)
        constexpr bool two = true;
        const void *fn = head_size == 128 ? (const void *)prefill_attention_flash<8, 2> : (const void *)prefill_attention_flash<4, 2>;
        if (lds_f > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f);
            if (e != hipSuccess) return e;
        }
        const dim3 grid(n_heads, (P + 63) / 64);
        if (x3 != nullptr && (kp & 3)) x3 = nullptr;
        void *params[] = {(void *)&q, (void *)&ldq, (void *)&kcache, (void *)&vcache, (void *)&out, (void *)&ldo,
                          (void *)&pos0, (void *)&P, (void *)&kv_dim, (void *)&kv_head, (void *)&kv_mul, (void *)&seq_len,
                          (void *)&x3, (void *)&kp};
        if (planes_written) *planes_written = x3 != nullptr;
        return hipLaunchKernel(fn, grid, dim3(two ? 512 : 256), params, lds_f, st);
    }
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
                               out, ldo, pos0, P, head_size, kv_dim, kv_head, kv_mul, seq_len);
        else if (tpw == 2)
            hipLaunchKernelGGL(prefill_attention_tiled<2>, grid, dim3(kPfBlock), lds_t, st, q, ldq, kcache, vcache,
                               out, ldo, pos0, P, head_size, kv_dim, kv_head, kv_mul, seq_len);
        else
            hipLaunchKernelGGL(prefill_attention_tiled<4>, grid, dim3(kPfBlock), lds_t, st, q, ldq, kcache, vcache,
                               out, ldo, pos0, P, head_size, kv_dim, kv_head, kv_mul, seq_len);
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
                       vcache, out, ldo, pos0, head_size, kv_dim, kv_head, kv_mul, seq_len);
    return hipGetLastError();
}

}  // namespace l2z
