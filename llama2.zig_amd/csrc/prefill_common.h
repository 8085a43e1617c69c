// prefill_common.h -- shared by the translation units of the batched prompt pass (prefill_gemm.hip,
// prefill_skinny.hip, prefill_attention.hip; host side: prefill_host.cpp through l2z_internal.h).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "l2z_internal.h"
#include "tunables.h"

namespace l2z {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int kPfBlock = 256;

enum GemmEpi { G_STORE = 0, G_RESID = 1, G_ROPE = 2, G_ROPE_CACHE = 3, G_CACHE = 4, G_SWIGLU = 5,
               G_QKV = 6,    // q | k | v in one launch: the epilogue of the block's column range (direct-to-LDS tile kernel only)
               G_SWIGLU_IL = 7 };  // W1 | W3 as ONE matrix of alternating rows (the device blob's slot: DESIGN.md 2): feature 2 p is
                                   // W1's row p, 2 p + 1 W3's -- adjacent lanes; out[token][p] = silu(a) * b (stream kernel only)

// main.zig:411-416 on the W3 product: out holds W1 x, becomes silu(W1 x) * (W3 x)
__device__ __forceinline__ float swiglu_merge(float h1, float h3)
{
    const float v = h1 * (1.0f / (1.0f + expf(-h1)));
    return v * h3;
}

struct GemmArgs {
    const float *x;      // [P, K] row-major (ldx floats per row)
    const float *w2;     // paired form only: the second [N, K] matrix (W3 beside W1)
    const float *w;      // [N, K] row-major
    float *out;          // [P, ldo]; G_*CACHE: cache base, row = pos0 + token
    const float *res;    // G_RESID: out = res + product ([P, ldres]; the unsharded pass has res == out)
    int P, N, K, ldx, ldo, ldres;
    int pos0;            // position of token 0 (RoPE angle, cache row)
    const float2 *rope;  // (seq_len, head_size/2) {cos, sin}
    int head_size;
    int n_scale;         // ranks the matrix's rows are sharded over (N * n_scale rows in the whole model)
    // G_QKV: features [0, nq) are rows of w (RoPE, out[token][f], ldo), [nq, nq + nkv) rows of wk (RoPE,
    // key-cache row pos0 + token, ldkv), the last nkv rows of wv (value-cache row); N = nq + 2 nkv
    const float *wk, *wv;
    float *outk, *outv;
    int nq, nkv, ldkv;
    // cache epilogues: != 0: the key / value caches are head-major [kv_heads][seq_len][head_size] and this is
    // seq_len * head_size; 0: flat rows of ldkv (ldo) floats
    size_t kv_head_stride;
    // direct-to-LDS tile kernel, 1-D grids (dma_grid): feature tiles, token tiles
    int ntx, nty;
    // split-K family of the direct-to-LDS tile kernel (sk = 2 or 4; 0 / 1: none): the sk blocks of a tile each
    // multiply a contiguous K / sk range, leave their accumulators in sk_part and bump the tile's counter in
    // sk_cnt; whichever arrives last adds the sk partials IN RANGE ORDER and runs the epilogue
    float *sk_part;
    int *sk_cnt;
    int sk;
    // floats between consecutive rows of w (and of w2): K for a plain [N, K] matrix, 2 K for W1 / W3, whose rows
    // alternate in one slot of the device blob.  Set by the launchers (0 is never valid).
    int ldw;
    // planes form of the tile kernel (X3): the activations as three planes of bf16 terms, x3[token][plane][kp]
    // (launch_split3 writes it from x just before the launch); ldx3 = 3 kp bf16 per token, kp = K rounded up to 64
    const void *x3;
    int ldx3, kp;
    // G_SWIGLU_IL (stream form): != null: the gated values' planes too, x3_out[token][plane][kp_out] (the next launch's x3)
    void *x3_out;
    int kp_out;
    // G_RESID (stream form, sk > 1): != 0: the blocks leave their K range's sums in sk_part and END -- the next rmsnorm
    // launch adds them and the residual (l2z_internal.h DeferredSum)
    int defer;
};


// where feature f of position pos lives in a cache whose flat form has rows of ld floats
__device__ __forceinline__ size_t kv_index(const GemmArgs &a, int ld, int pos, int f)
{
    return a.kv_head_stride ? (size_t)(f / a.head_size) * a.kv_head_stride + (size_t)pos * (size_t)a.head_size +
                                  (size_t)(f % a.head_size)
                            : (size_t)pos * (size_t)ld + (size_t)f;
}

// One direct-to-LDS load: 16 bytes per lane from `g` (per lane) to lds + 16 * lane (`lds` wave-uniform).
// A plain __device__ function: called from the kernel TEMPLATE directly, the builtin makes the
// host-side instantiation fail silently (no launch stub is emitted, the library then does not link).
__device__ __forceinline__ void lds_dma16(const float *g, float *lds)
{
    __builtin_amdgcn_global_load_lds(g, lds, 16, 0, 0);
}
// the same with the non-temporal policy (aux = 2): a stream that one CU reads once
__device__ __forceinline__ void lds_dma16_nt(const float *g, float *lds)
{
    __builtin_amdgcn_global_load_lds(g, lds, 16, 0, 2);
}

// ---- f32 products on the bf16 matrix cores (round 6): x = x1 + x2 + x3, three bf16 terms ----
// A float is split into three bf16 terms, each the round-to-nearest-even bf16 of what the terms before it left
// (x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2): 3 x 8 significand bits, the subtractions exact, so
// x1 + x2 + x3 == x for every finite float whose last term does not underflow).  A product x w is then the sum of
// nine bf16 x bf16 products, each EXACT in f32; the six of order <= 2^-16 are summed into the f32 accumulator by
// v_mfma_f32_32x32x16_bf16 (the three left out are <= 2^-24 |x w| each: below the rounding of ONE f32 product).
// The matrix cores run bf16 at 16 x the f32 rate: six instructions of 32 cycles replace eight of 64 per 16 k.
#if defined(__HIPCC__)
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
struct Bf3 { v8bf t1, t2, t3; };

// eight consecutive floats of a row (two float4) -> their three bf16 terms, element e of each = float e
__device__ __forceinline__ Bf3 split3(const v4f lo, const v4f hi)
{
    Bf3 o;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const v2f f = {e < 4 ? lo[e] : hi[e - 4], e < 4 ? lo[e + 1] : hi[e - 3]};
        const v2bf p1 = __builtin_convertvector(f, v2bf);           // v_cvt_pk_bf16_f32: round to nearest even
        const v2f r1 = f - __builtin_convertvector(p1, v2f);        // exact
        const v2bf p2 = __builtin_convertvector(r1, v2bf);
        const v2f r2 = r1 - __builtin_convertvector(p2, v2f);       // exact
        const v2bf p3 = __builtin_convertvector(r2, v2bf);
        o.t1[e] = p1[0]; o.t1[e + 1] = p1[1];
        o.t2[e] = p2[0]; o.t2[e + 1] = p2[1];
        o.t3[e] = p3[0]; o.t3[e + 1] = p3[1];
    }
    return o;
}

// the same terms of FOUR consecutive floats (the producers of an activation matrix write its planes beside it: rmsnorm,
// the attention output -- an element's terms depend on that element alone, so these are the bits split3 gives)
typedef __bf16 v4bf __attribute__((ext_vector_type(4)));
struct Bf3x4 { v4bf t1, t2, t3; };
__device__ __forceinline__ Bf3x4 split3_4(const v4f x)
{
    Bf3x4 o;
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
        const v2f f = {x[e], x[e + 1]};
        const v2bf p1 = __builtin_convertvector(f, v2bf);
        const v2f r1 = f - __builtin_convertvector(p1, v2f);
        const v2bf p2 = __builtin_convertvector(r1, v2bf);
        const v2f r2 = r1 - __builtin_convertvector(p2, v2f);
        const v2bf p3 = __builtin_convertvector(r2, v2bf);
        o.t1[e] = p1[0]; o.t1[e + 1] = p1[1];
        o.t2[e] = p2[0]; o.t2[e + 1] = p2[1];
        o.t3[e] = p3[0]; o.t3[e + 1] = p3[1];
    }
    return o;
}
// ... and of one float
__device__ __forceinline__ void split3_1(float x, __bf16 &t1, __bf16 &t2, __bf16 &t3)
{
    t1 = (__bf16)x;
    const float r1 = x - (float)t1;
    t2 = (__bf16)r1;
    const float r2 = r1 - (float)t2;
    t3 = (__bf16)r2;
}
// four consecutive elements k .. k + 3 of token `tok` into the planes matrix x3[token][plane][kp] (k % 4 == 0)
__device__ __forceinline__ void planes_store4(__bf16 *x3, int kp, int tok, int k, const v4f v)
{
    const Bf3x4 t = split3_4(v);
    __bf16 *o = x3 + (size_t)tok * 3 * kp + k;
    *(v4bf *)o = t.t1; *(v4bf *)(o + kp) = t.t2; *(v4bf *)(o + 2 * kp) = t.t3;
}
__device__ __forceinline__ void planes_store1(__bf16 *x3, int kp, int tok, int k, float v)
{
    __bf16 a, b, c;
    split3_1(v, a, b, c);
    __bf16 *o = x3 + (size_t)tok * 3 * kp + k;
    o[0] = a; o[kp] = b; o[2 * kp] = c;
}

// acc += sum over the lanes' 16 k of a b, smallest terms first (one fixed order: part of the arithmetic)
__device__ __forceinline__ v16f x3_mfma(const Bf3 &a, const Bf3 &b, v16f acc)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t3, b.t1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t1, b.t3, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t2, b.t2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t2, b.t1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t1, b.t2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t1, b.t1, acc, 0, 0, 0);
    return acc;
}
#endif

// Loop bound of a kernel that multiplies whole `granule`-k stages (round 6): the product's K rounded up.  The columns
// past K are ZEROS in every activation matrix of the batched pass (prefill_host.cpp pf_ld: rows padded to a multiple of
// 256 floats, >= 768), and whatever follows the row in W -- the next row, the next tensor, the zeroed slack behind the
// blob (weights.cpp kBlobSlackFloats) -- is finite: the extra products are exact zeros.  -1: the activation rows are
// not padded that far (a caller that did not come through prefill_host.cpp).
inline int pad_k(int K, int granule, int ldx)
{
    const int ke = (K + granule - 1) / granule * granule;
    return ke <= ldx ? ke : -1;
}

// prefill_skinny.hip: the short-prompt (P <= 64 tokens) GEMM forms; picks the form and the token tiling
hipError_t launch_prefill_skinny(int epi, const GemmArgs &a, hipStream_t st);
hipError_t launch_prefill_skinny_pair(int epi, const GemmArgs &a, hipStream_t st);  // G_SWIGLU: w | w2 gated; G_QKV: wk | wv


}  // namespace l2z
