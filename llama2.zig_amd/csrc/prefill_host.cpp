// prefill_host.cpp -- host side of the batched prompt prefill (kernels: prefill_gemm.hip, prefill_skinny.hip, prefill_attention.hip).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "l2z_state.h"

using namespace l2z;

// ---------------------------------------------------------------------------
// Batched prefill: the same state change as calling
// l2z_transformer(tokens[i], pos0 + i) for i = 0..n-1 -- KV-cache rows pos0..pos0+n-1 written in
// every layer, logits of the LAST position left in the runstate -- but every weight matrix is
// streamed once per chunk of up to kPrefillChunk tokens and multiplied on the fp32 matrix cores.
namespace {
#define kPrefillChunk prefill_chunk_tokens()

// Row pitch of the activation matrices the GEMMs read (pf_xn, pf_att, pf_h1): the width rounded up to a multiple of 256
// floats, at least 768, the pad columns ZERO and never written.  Every GEMM kernel of the pass multiplies whole stages of
// K (64 / 128 / 256 floats): it runs over K rounded up (pad_k), reads zeros from these rows and, past the end of a W row,
// whatever finite floats follow -- so any K that is a multiple of 4 takes the direct-to-LDS kernels (until round 5
// K % 64 != 0 fell back to register-staged forms: stories15M's dim 288, stories42M's hidden_dim 1376).
int pf_ld(int n)
{
    const int r = (n + 255) / 256 * 256;
    return r < 768 ? 768 : r;
}

int prefill_alloc(l2z_runstate *s, int need)
{
    const l2z_config &c = s->cfg;
    if (s->pf_cap >= need) return L2Z_OK;
    // first use, or a longer chunk than any before (a prompt of >= 1024 tokens; L2Z_PF_CHUNK through
    // l2z_option_set): start over, in steps of 512 tokens
    const size_t P = (size_t)(need + 511) / 512 * 512;
    L2Z_HIP(hipStreamSynchronize(s->stream));
    float **bufs[] = {&s->pf_x, &s->pf_xn, &s->pf_q, &s->pf_att, &s->pf_h1, &s->pf_stage, &s->pf_part};
    for (float **b : bufs)
        if (*b) { (void)hipFree(*b); *b = nullptr; }
    if (s->pf_tokens) { (void)hipFree(s->pf_tokens); s->pf_tokens = nullptr; }
    if (s->pf_sk.x3) { (void)hipFree(s->pf_sk.x3); s->pf_sk.x3 = nullptr; s->pf_sk.x3_bytes = 0; }
    if (s->pf_sk.x3b) { (void)hipFree(s->pf_sk.x3b); s->pf_sk.x3b = nullptr; s->pf_sk.x3b_bytes = 0; }
    s->pf_cap = 0;
    const size_t widest = (size_t)(c.dim > c.hidden_dim ? c.dim : c.hidden_dim);
    // (scheme B reads the local attention / hidden blocks as rows of the column shards' PADDED width: a 1-rank group's
    // padded width can exceed the model's)
    const size_t att_w = (size_t)pf_ld(std::max(c.dim, s->sh.scheme_b ? s->sh.dimc_pad : 0));
    const size_t h1_w = (size_t)pf_ld(std::max(c.hidden_dim, s->sh.scheme_b ? s->sh.hidc_pad : 0));
    const size_t xn_w = (size_t)pf_ld(c.dim);
    s->pf_ld_xn = (int)xn_w; s->pf_ld_att = (int)att_w; s->pf_ld_h1 = (int)h1_w;
    struct { void **p; size_t bytes; } want[] = {
        {(void **)&s->pf_x, P * c.dim * 4},   {(void **)&s->pf_xn, P * xn_w * 4},
        {(void **)&s->pf_q, P * c.dim * 4},   {(void **)&s->pf_att, P * att_w * 4},
        {(void **)&s->pf_h1, P * h1_w * 4},
        // sharded: [world][P, n / world] blocks of the matrix being gathered
        {(void **)&s->pf_stage, s->sh.world > 1 ? P * widest * 4 : 0},
        // scheme B: this rank's partial [P, dim] products of its column shards of Wo / W2 (summed by the bulk all-reduce)
        {(void **)&s->pf_part, s->sh.scheme_b ? P * c.dim * 4 : 0},
        {(void **)&s->pf_tokens, P * 4},
        // the tile GEMM on the bf16 matrix cores: one launch's activation matrix as three planes of bf16 terms
        {(void **)&s->pf_sk.x3, P * 3 * std::max(std::max(xn_w, att_w), h1_w) * 2},
        // ... and the gated hidden rows' planes, written by the W1 | W3 launch while it reads its own (unsharded pass)
        {(void **)&s->pf_sk.x3b, s->sh.world > 1 ? 0 : P * 3 * h1_w * 2}};
    for (auto &b : want) {
        if (b.bytes == 0) continue;
        hipError_t e = hipMalloc(b.p, b.bytes);
        if (e != hipSuccess) {
            *b.p = nullptr;
            set_error("prefill scratch allocation (%zu bytes) failed: %s", b.bytes, hipGetErrorString(e));
            return e == hipErrorOutOfMemory ? L2Z_ERR_OOM : L2Z_ERR_HIP;  // what was allocated is freed with the runstate, or on the next attempt
        }
    }
    s->pf_cap = (int)P;
    s->pf_sk.x3_bytes = P * 3 * std::max(std::max(xn_w, att_w), h1_w) * 2;
    s->pf_sk.x3b_bytes = s->sh.world > 1 ? 0 : P * 3 * h1_w * 2;
    // the pad columns are never written and must be zeros: the GEMMs multiply them against whatever follows a W row
    // (scheme B: against the column shards' own zero columns)
    L2Z_HIP(hipMemsetAsync(s->pf_xn, 0, P * xn_w * 4, s->stream));
    L2Z_HIP(hipMemsetAsync(s->pf_att, 0, P * att_w * 4, s->stream));
    L2Z_HIP(hipMemsetAsync(s->pf_h1, 0, P * h1_w * 4, s->stream));
    if (s->pf_sk.part == nullptr) {
        // split-K family of the tile GEMM (chunks of 33 ... 256 tokens): accumulator dumps of up to 4 K ranges of
        // the widest launch of a layer (q | k | v, or W1 | W3 side by side), one arrival counter per output tile
        const Shard &sh = s->sh;
        const size_t widest_launch = std::max((size_t)sh.dim_loc + 2 * (size_t)sh.kvd_loc, 2 * (size_t)sh.hid_loc) + 128;
        const size_t floats = (size_t)kPanelWsRows * widest_launch;  // (the split-K family needs 4 kSplitKMaxTokens rows of it)
        const int n_cnt = 1 << 16;
        hipError_t e = hipMalloc((void **)&s->pf_sk.part, floats * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&s->pf_sk.cnt, (size_t)n_cnt * 4);
        if (e == hipSuccess) e = hipMemset(s->pf_sk.cnt, 0, (size_t)n_cnt * 4);
        if (e != hipSuccess) {
            set_error("prefill split-K workspace allocation failed: %s", hipGetErrorString(e));
            return e == hipErrorOutOfMemory ? L2Z_ERR_OOM : L2Z_ERR_HIP;
        }
        s->pf_sk.part_floats = floats;
        s->pf_sk.cnt_ints = n_cnt;
    }
    return L2Z_OK;
}

// A layer is four stages, each ending in a [P, n] matrix whose columns are split over the ranks
// exactly like the decode pass's vectors (forward.cpp): attention output by heads, the two
// residual updates by rows of wo / w2, the gated hidden row by rows of w1 / w3.
enum { PF_ATT = 0, PF_WO, PF_H1, PF_W2, PF_STAGES };

struct StageOut {
    float *dst;  // the row-major [P, ldd] matrix the next stage reads
    int n_loc, ldd;
};
StageOut stage_out(const l2z_runstate *s, int k)
{
    const l2z_config &c = s->cfg;
    const Shard &sh = s->sh;
    switch (k) {
        case PF_ATT: return {s->pf_att, sh.dim_loc, s->pf_ld_att};
        case PF_H1: return {s->pf_h1, sh.hid_loc, s->pf_ld_h1};
        default: return {s->pf_x, sh.dim_loc, c.dim};  // PF_WO, PF_W2
    }
}

// The launches of stage k of layer l.  One rank: straight into the destination matrix.  Sharded: this
// rank's [P, n_loc] block, contiguous, at pf_stage + rank * P * n_loc; the exchange and the unpack
// into the destination follow (comm_bulk_allgather, or the emulated-rank driver's copies).
int prefill_stage(l2z_runstate *s, const l2z_weights *w, int l, int k, int P, int pos0)
{
    const l2z_config &c = s->cfg;
    const Shard &sh = s->sh;
    hipStream_t st = s->stream;
    const int dim = c.dim, hid = c.hidden_dim, kvd = sh.kvd_loc, hs = sh.hs;
    const bool sharded = sh.world > 1;
    const StageOut o = stage_out(s, k);
    float *out = sharded ? s->pf_stage + (size_t)sh.rank * P * o.n_loc : o.dst;
    const int ldo = sharded ? o.n_loc : o.ldd;
    float *kc = s->key_cache + (size_t)l * c.seq_len * kvd;  // this layer: [kv heads][seq_len][hs]
    float *vc = s->value_cache + (size_t)l * c.seq_len * kvd;
    const size_t kvh_stride = (size_t)c.seq_len * hs;
    // K ranges per output tile (split-K family, prefill_gemm.hip): part of the arithmetic, so taken from the WHOLE
    // model's matrices -- q | k | v as one launch's width whether or not they go out as one launch
    const int kvd_whole = c.n_kv_heads * hs;
    const int ldxn = s->pf_ld_xn, ldatt = s->pf_ld_att, ldh1 = s->pf_ld_h1;   // padded rows of the GEMM inputs (pf_ld)
    const int dim64 = (dim + 63) / 64 * 64, hid64 = (hid + 63) / 64 * 64;     // K as the tile GEMM walks it
    const int sk_qkv = prefill_split_k((long long)dim + 2 * kvd_whole, P, dim64, false);
    const int sk_wo = prefill_split_k(dim, P, dim64, false), sk_w2 = prefill_split_k(dim, P, hid64, false);
    const int sk_h1 = prefill_split_k(hid, P, dim64, true);
    const SplitKWs *ws = &s->pf_sk;
    // Chunks of <= 32 tokens of matrices that stream from HBM: the K-range panel kernel (prefill_panel.hip), chosen from
    // the WHOLE model's matrix so that a shard takes what the unsharded pass takes.
    auto panel = [&](PanelProduct &pp, long long n_whole, bool *taken) -> int {
        *taken = false;
        const long long widest_whole = std::max((long long)dim + 2 * kvd_whole, 2LL * hid) + 128;
        if (!prefill_panel_shape(n_whole, P, pp.K, widest_whole)) return L2Z_OK;
        pp.P = P;
        const hipError_t e = launch_prefill_panel(pp, g_cus, ws, st);
        if (e == hipErrorNotSupported) {
            // The WHOLE model's product takes the panel kernel but this launch cannot (rows not a multiple of 16,
            // alignment, workspace).  Unsharded: the forms below, consistently.  On a shard the unsharded pass takes
            // the panel kernel's summation order and this rank would take another: refuse rather than break
            // "row-sharded == unsharded, bit for bit" silently (prefill_usable keeps such shards off this path).
            L2Z_CHECK(!sharded, L2Z_ERR_INVALID,
                      "batched prefill: this rank's share of a [%lld, %d] product does not take the kernel the unsharded "
                      "pass takes (rows per rank must be multiples of 16)", n_whole, pp.K);
            return L2Z_OK;
        }
        L2Z_HIP(e);
        *taken = true;
        return L2Z_OK;
    };
    bool taken = false;
    // The unsharded pass: the PRODUCER of an activation matrix writes its planes of bf16 terms beside it where the consumer
    // multiplies on the bf16 matrix cores (rmsnorm -> q | k | v and W1 | W3; the attention output -> Wo; the SwiGLU epilogue
    // -> W2), so the consumer's split launch (prefill_gemm.hip prepare_x3) is not needed: four launches less per layer.
    // The same bits as the split launch's (an element's terms depend on that element alone).  Rows whose K is not a whole
    // number of 64-k stages keep the split launch (it writes the pad columns' zeros); a shard's operands arrive through
    // the exchange and keep it too.  planes_for: whether a [P, K] x [n_whole, K]^T product reads planes at all.
    auto planes_for = [&](long long n_whole, int K, int sk) {
        const int kp = (K + 63) / 64 * 64;
        return !sharded && tunables().pf_fuse_planes != 0 && kp == K && x3_applies(n_whole, kp) &&
               (x3_stream_shape(n_whole, P, kp) || P > Tunables::pf_skinny_max || sk > 1) && ws->x3 != nullptr &&
               (size_t)P * 3 * kp * 2 <= ws->x3_bytes;
    };
    int xn_planes = PLANES_SPLIT;
    if (k == PF_ATT || k == PF_H1) {  // :305 / :398
        const bool pl = k == PF_ATT ? planes_for((long long)dim + 2 * kvd_whole, dim, sk_qkv) : planes_for(2LL * hid, dim, sk_h1);
        // (... and adds the K ranges' sums the residual product before it left behind: DeferredSum)
        L2Z_HIP(launch_prefill_rmsnorm(s->pf_xn, ldxn, s->pf_x, (k == PF_ATT ? w->rms_att : w->rms_ffn) + (size_t)l * dim, dim, P, st,
                                       pl ? ws->x3 : nullptr, dim, &s->pf_pending));
        s->pf_pending.valid = false;
        if (pl) xn_planes = PLANES_READY;
    }
    L2Z_CHECK(!s->pf_pending.valid, L2Z_ERR_INVALID, "batched prefill: a residual product's sums were left for a norm that did not run (stage %d)", k);
    // Wo / W2 may leave their K ranges' sums to the rmsnorm launch that reads the residual stream next (the unsharded pass; W2
    // of the last layer finishes itself: the classifier reads x)
    const bool may_defer = !sharded && tunables().pf_fuse_planes != 0;
    if (k == PF_ATT) {
        {
            PanelProduct pp = {};
            pp.x = s->pf_xn; pp.ldx = ldxn; pp.K = dim;
            pp.w0 = w->wq + (size_t)l * sh.dim_loc * dim; pp.w1 = w->wk + (size_t)l * kvd * dim; pp.w2 = w->wv + (size_t)l * kvd * dim;
            pp.rows0 = sh.dim_loc; pp.rows1 = kvd; pp.rows2 = kvd;
            pp.mode = PANEL_QKV; pp.out = s->pf_q; pp.ldo = sh.dim_loc; pp.outk = kc; pp.outv = vc; pp.ldkv = kvd;
            pp.head_size = hs; pp.pos0 = pos0; pp.kv_head_stride = kvh_stride; pp.rope = s->rope;
            L2Z_TRY(panel(pp, (long long)dim + 2 * kvd_whole, &taken));
        }
        if (!taken) {
        // q of the local heads ([P, dim_loc]) and the k / v rows of the local kv heads: one launch where
        // the tile kernel takes the shape (:308-358), else three
        const float *wq = w->wq + (size_t)l * sh.dim_loc * dim, *wk = w->wk + (size_t)l * kvd * dim,
                    *wv = w->wv + (size_t)l * kvd * dim;
        const hipError_t qe = launch_prefill_gemm_qkv(s->pf_xn, ldxn, wq, wk, wv, s->pf_q, sh.dim_loc, kc, vc, kvd, P,
                                                      sh.dim_loc, kvd, dim, pos0, s->rope, hs, st, kvh_stride, sh.world,
                                                      sk_qkv, ws, xn_planes);
        if (qe == hipErrorNotSupported) {
            const long long n_qkv = (long long)dim + 2 * kvd_whole;   // the whole model's q | k | v launch (the stream form's K ranges)
            L2Z_HIP(launch_prefill_gemm(PG_ROPE, s->pf_xn, ldxn, wq, s->pf_q, sh.dim_loc, P, sh.dim_loc, dim, pos0,
                                        s->rope, hs, st, nullptr, 0, sh.world, 0, sk_qkv, ws, 0, n_qkv, xn_planes));  // :308-351
            const hipError_t ke = launch_prefill_gemm_kv_pair(s->pf_xn, ldxn, wk, wv, kc, vc, kvd, P, kvd, dim, pos0,
                                                              s->rope, hs, st, sh.world, kvh_stride, sk_qkv, n_qkv);  // short prompts: k | v together
            if (ke == hipErrorNotSupported) {
                L2Z_HIP(launch_prefill_gemm(PG_ROPE_CACHE, s->pf_xn, ldxn, wk, kc, kvd, P, kvd, dim, pos0, s->rope, hs,
                                            st, nullptr, 0, sh.world, kvh_stride, sk_qkv, ws, 0, n_qkv, PLANES_READY));   // :354-357 (q's planes stand)
                L2Z_HIP(launch_prefill_gemm(PG_CACHE, s->pf_xn, ldxn, wv, vc, kvd, P, kvd, dim, pos0, s->rope, hs, st,
                                            nullptr, 0, sh.world, kvh_stride, sk_qkv, ws, 0, n_qkv, PLANES_READY));       // :358
            } else {
                L2Z_HIP(ke);
            }
        } else {
            L2Z_HIP(qe);
        }
        }
        // (the attention output's planes for the Wo product: nothing reads ws->x3 any more -- q | k | v are done)
        bool att_planes = false;
        const bool want = planes_for(dim, dim, sk_wo);
        L2Z_HIP(launch_prefill_attention(s->pf_q, sh.dim_loc, kc, vc, out, ldo, pos0, P, sh.heads_loc, hs,
                                         hs, kvh_stride, c.n_heads / c.n_kv_heads, c.seq_len, st, c.n_heads, 0,
                                         want ? ws->x3 : nullptr, dim, &att_planes));  // :361-389
        s->pf_planes_att = att_planes ? PLANES_READY : PLANES_SPLIT;
    } else if (k == PF_WO) {
        const float *res = s->pf_x + sh.dim0;
        {
            PanelProduct pp = {};
            pp.x = s->pf_att; pp.ldx = ldatt; pp.K = dim; pp.w0 = w->wo + (size_t)l * sh.dim_loc * dim; pp.rows0 = sh.dim_loc;
            pp.mode = PANEL_RESID; pp.out = out; pp.ldo = ldo; pp.res = res; pp.ldres = dim;
            L2Z_TRY(panel(pp, dim, &taken));
        }
        if (!taken)
        L2Z_HIP(launch_prefill_gemm(PG_RESID, s->pf_att, ldatt, w->wo + (size_t)l * sh.dim_loc * dim, out, ldo,
                                    P, sh.dim_loc, dim, pos0, s->rope, hs, st, res, dim, sh.world, 0, sk_wo, ws, 0, 0,
                                    sharded ? PLANES_SPLIT : s->pf_planes_att, may_defer ? &s->pf_pending : nullptr));   // :392-395
        s->pf_planes_att = PLANES_SPLIT;
    } else if (k == PF_H1) {
        // :405-416: W1 and W3 in one launch with silu(a) * b as its epilogue where the tile kernel
        // takes the shape, else two GEMMs, the second one merging into the first one's output
        // W1 | W3 share one slot of the device blob, rows alternating: rows of either are 2 dim apart
        const float *w1 = w->w1 + (size_t)l * sh.hid_loc * 2 * dim, *w3 = w->w3 + (size_t)l * sh.hid_loc * 2 * dim;
        {
            PanelProduct pp = {};   // the shared slot as ONE matrix of 2 hid_loc rows: row 2p = W1 row p, 2p + 1 = W3 row p
            pp.x = s->pf_xn; pp.ldx = ldxn; pp.K = dim; pp.w0 = w1; pp.rows0 = 2 * sh.hid_loc;
            pp.mode = PANEL_SWIGLU; pp.out = out; pp.ldo = ldo;
            L2Z_TRY(panel(pp, 2LL * hid, &taken));
        }
        s->pf_planes_h1 = PLANES_SPLIT;
        if (taken) return L2Z_OK;
        bool h1_planes = false;
        const hipError_t pe = launch_prefill_gemm_swiglu_pair(s->pf_xn, ldxn, w1, w3, out, ldo, P, sh.hid_loc, dim, st, sh.world,
                                                              sk_h1, ws, 2 * dim, xn_planes,
                                                              planes_for(dim, hid, sk_w2) ? hid : 0, &h1_planes);
        if (pe == hipSuccess && h1_planes) s->pf_planes_h1 = PLANES_READY_B;
        if (pe == hipErrorNotSupported) {
            L2Z_HIP(launch_prefill_gemm(PG_STORE, s->pf_xn, ldxn, w1, out, ldo, P, sh.hid_loc, dim, pos0, s->rope,
                                        hs, st, nullptr, 0, sh.world, 0, sk_h1, ws, 2 * dim, 0, xn_planes));                      // :405
            L2Z_HIP(launch_prefill_gemm(PG_SWIGLU, s->pf_xn, ldxn, w3, out, ldo, P, sh.hid_loc, dim, pos0, s->rope,
                                        hs, st, nullptr, 0, sh.world, 0, sk_h1, ws, 2 * dim, 0,
                                        xn_planes));  // :408 + :411-416 in the epilogue (W1's planes stand -- or rmsnorm's)
        } else {
            L2Z_HIP(pe);
        }
    } else {
        const float *res = s->pf_x + sh.dim0;
        {
            PanelProduct pp = {};
            pp.x = s->pf_h1; pp.ldx = ldh1; pp.K = hid; pp.w0 = w->w2 + (size_t)l * sh.dim_loc * hid; pp.rows0 = sh.dim_loc;
            pp.mode = PANEL_RESID; pp.out = out; pp.ldo = ldo; pp.res = res; pp.ldres = dim;
            L2Z_TRY(panel(pp, dim, &taken));
        }
        if (!taken)
        L2Z_HIP(launch_prefill_gemm(PG_RESID, s->pf_h1, ldh1, w->w2 + (size_t)l * sh.dim_loc * hid, out, ldo,
                                    P, sh.dim_loc, hid, pos0, s->rope, hs, st, res, dim, sh.world, 0, sk_w2, ws, 0, 0,
                                    sharded ? PLANES_SPLIT : s->pf_planes_h1,
                                    may_defer && l + 1 < c.n_layers ? &s->pf_pending : nullptr));   // :419-422
        s->pf_planes_h1 = PLANES_SPLIT;
    }
    return L2Z_OK;
}

// Scheme B (L2Z_SCHEME_B: Wo and W2 sharded by COLUMNS; forward.cpp): a layer is two halves, each ending in a partial
// [P, dim] product of this rank's column shard -- pf_part, rank 0's with the residual (main.zig:395 / :422) -- that the
// bulk all-reduce sums over the ranks in rank order into pf_x.  The attention output and the gated hidden rows of the
// rank's own heads / hidden rows never leave it: they are the K-slices the column shards multiply.
//   half 0: rmsnorm, q | k | v of the local heads, attention -> pf_att [P, dimc_pad]; Wo columns -> pf_part
//   half 1: rmsnorm, W1 | W3 of the local rows -> pf_h1 [P, hidc_pad]; W2 columns -> pf_part
// The products are split across ranks differently than in the unsharded pass: logits at the parity tolerance, the ranks
// bit-identical to each other (tests/test_gpu_scheme_b.py).
int prefill_half_b(l2z_runstate *s, const l2z_weights *w, int l, int half, int P, int pos0)
{
    const l2z_config &c = s->cfg;
    const Shard &sh = s->sh;
    hipStream_t st = s->stream;
    const int dim = c.dim, kvd = sh.kvd_loc, hs = sh.hs;
    float *kc = s->key_cache + (size_t)l * c.seq_len * kvd;
    float *vc = s->value_cache + (size_t)l * c.seq_len * kvd;
    const size_t kvh_stride = (size_t)c.seq_len * hs;
    const SplitKWs *ws = &s->pf_sk;
    // kernel forms are chosen from what every rank sees alike: the whole matrices' row counts and the shards' widths
    const int kvd_whole = c.n_kv_heads * hs;
    const int epi = sh.rank == 0 ? PG_RESID : PG_STORE;
    const int ldxn = s->pf_ld_xn, ldatt = s->pf_ld_att, ldh1 = s->pf_ld_h1;   // padded rows of the GEMM inputs (pf_ld)
    const int dim64 = (dim + 63) / 64 * 64;
    if (half == 0) {
        L2Z_HIP(launch_prefill_rmsnorm(s->pf_xn, ldxn, s->pf_x, w->rms_att + (size_t)l * dim, dim, P, st));  // :305
        const int sk_qkv = prefill_split_k((long long)dim + 2 * kvd_whole, P, dim64, false);
        const float *wq = w->wq + (size_t)l * sh.dim_loc * dim, *wk = w->wk + (size_t)l * kvd * dim, *wv = w->wv + (size_t)l * kvd * dim;
        const hipError_t qe = launch_prefill_gemm_qkv(s->pf_xn, ldxn, wq, wk, wv, s->pf_q, sh.dim_loc, kc, vc, kvd, P, sh.dim_loc, kvd,
                                                      dim, pos0, s->rope, hs, st, kvh_stride, sh.world, sk_qkv, ws);
        if (qe == hipErrorNotSupported) {
            const long long n_qkv = (long long)dim + 2 * kvd_whole;
            L2Z_HIP(launch_prefill_gemm(PG_ROPE, s->pf_xn, ldxn, wq, s->pf_q, sh.dim_loc, P, sh.dim_loc, dim, pos0, s->rope, hs, st,
                                        nullptr, 0, sh.world, 0, sk_qkv, ws, 0, n_qkv));
            L2Z_HIP(launch_prefill_gemm(PG_ROPE_CACHE, s->pf_xn, ldxn, wk, kc, kvd, P, kvd, dim, pos0, s->rope, hs, st, nullptr, 0,
                                        sh.world, kvh_stride, sk_qkv, ws, 0, n_qkv, true));
            L2Z_HIP(launch_prefill_gemm(PG_CACHE, s->pf_xn, ldxn, wv, vc, kvd, P, kvd, dim, pos0, s->rope, hs, st, nullptr, 0,
                                        sh.world, kvh_stride, sk_qkv, ws, 0, n_qkv, true));
        } else {
            L2Z_HIP(qe);
        }
        L2Z_HIP(launch_prefill_attention(s->pf_q, sh.dim_loc, kc, vc, s->pf_att, ldatt, pos0, P, sh.heads_loc, hs, hs,
                                         kvh_stride, c.n_heads / c.n_kv_heads, c.seq_len, st, c.n_heads));   // :361-389
        const int sk = prefill_split_k(dim, P, (sh.dimc_pad + 63) / 64 * 64, false);
        L2Z_HIP(launch_prefill_gemm(epi, s->pf_att, ldatt, w->wo + (size_t)l * dim * sh.dimc_pad, s->pf_part, dim, P, dim,
                                    sh.dimc_pad, pos0, s->rope, hs, st, s->pf_x, dim, 1, 0, sk, ws));        // :392-395
    } else {
        L2Z_HIP(launch_prefill_rmsnorm(s->pf_xn, ldxn, s->pf_x, w->rms_ffn + (size_t)l * dim, dim, P, st));  // :398
        const int sk_h1 = prefill_split_k(c.hidden_dim, P, dim64, true);
        const float *w1 = w->w1 + (size_t)l * sh.hid_loc * 2 * dim, *w3 = w->w3 + (size_t)l * sh.hid_loc * 2 * dim;
        const hipError_t pe = launch_prefill_gemm_swiglu_pair(s->pf_xn, ldxn, w1, w3, s->pf_h1, ldh1, P, sh.hid_loc, dim, st,
                                                              sh.world, sk_h1, ws, 2 * dim);
        if (pe == hipErrorNotSupported) {
            L2Z_HIP(launch_prefill_gemm(PG_STORE, s->pf_xn, ldxn, w1, s->pf_h1, ldh1, P, sh.hid_loc, dim, pos0, s->rope, hs, st,
                                        nullptr, 0, sh.world, 0, sk_h1, ws, 2 * dim));                     // :405
            L2Z_HIP(launch_prefill_gemm(PG_SWIGLU, s->pf_xn, ldxn, w3, s->pf_h1, ldh1, P, sh.hid_loc, dim, pos0, s->rope, hs,
                                        st, nullptr, 0, sh.world, 0, sk_h1, ws, 2 * dim));                 // :408-416
        } else {
            L2Z_HIP(pe);
        }
        const int sk = prefill_split_k(dim, P, (sh.hidc_pad + 63) / 64 * 64, false);
        L2Z_HIP(launch_prefill_gemm(epi, s->pf_h1, ldh1, w->w2 + (size_t)l * dim * sh.hidc_pad, s->pf_part, dim, P, dim,
                                    sh.hidc_pad, pos0, s->rope, hs, st, s->pf_x, dim, 1, 0, sk, ws));        // :419-422
    }
    return L2Z_OK;
}

int prefill_begin_chunk(l2z_runstate *s, const l2z_weights *w, const int32_t *tokens, int P)
{
    s->pf_pending.valid = false;
    // the split forms' arrival counters and flags are left at zero by every launch that completes; a pass that was
    // cut short (a peer-write wait that timed out, a failed launch) must not leave a later one a half-counted tile
    if (s->pf_sk.cnt)
        L2Z_HIP(hipMemsetAsync(s->pf_sk.cnt, 0, (size_t)s->pf_sk.cnt_ints * sizeof(int), s->stream));
    L2Z_HIP(hipMemcpyAsync(s->pf_tokens, tokens, (size_t)P * 4, hipMemcpyHostToDevice, s->stream));
    L2Z_HIP(launch_prefill_embed(s->pf_x, w->tok_emb, s->pf_tokens, s->cfg.dim, P, s->stream));  // :295
    return L2Z_OK;
}

int prefill_chunk(l2z_runstate *s, const l2z_weights *w, const int32_t *tokens, int P, int pos0)
{
    L2Z_TRY(prefill_begin_chunk(s, w, tokens, P));
    if (s->sh.scheme_b) {
        for (int l = 0; l < s->cfg.n_layers; l++)
            for (int half = 0; half < 2; half++) {
                L2Z_TRY(prefill_half_b(s, w, l, half, P, pos0));
                L2Z_TRY(comm_bulk_allreduce(s->comm, s->pf_part, P, s->cfg.dim, s->pf_stage, s->pf_x, s->cfg.dim, s->stream));
            }
        return L2Z_OK;
    }
    for (int l = 0; l < s->cfg.n_layers; l++)
        for (int k = 0; k < PF_STAGES; k++) {
            L2Z_TRY(prefill_stage(s, w, l, k, P, pos0));
            if (s->sh.world > 1) {
                const StageOut o = stage_out(s, k);
                L2Z_TRY(comm_bulk_allgather(s->comm, s->pf_stage, P, o.n_loc, o.dst, o.ldd, s->stream));
            }
        }
    return L2Z_OK;
}

// the last position's residual row is RunState.x: the usual final rmsnorm + classifier launch
// (:426-429) over this rank's vocabulary rows; sharded, the caller gathers the logits
int prefill_classifier(l2z_runstate *s, const l2z_weights *w)
{
    const l2z_config &c = s->cfg;
    const Shard &sh = s->sh;
    MatvecArgs a = {};
    a.w0 = w->wcls; a.out0 = s->logits + sh.v0; a.rows0 = sh.v_loc; a.n = c.dim; a.x = s->x;
    a.rms_w = w->rms_final;
    a.part_val = s->d_part_val; a.part_idx = s->d_part_idx; a.row_offset = sh.v0;
    int grid = 0;
    const bool fuse = sh.world == 1 && matvec_vector_width(c.dim);  // as in enqueue_forward
    L2Z_HIP(launch_matvec(a, PRO_RMS, fuse ? EPI_ARGMAX : EPI_STORE, s->max_blocks, g_cus, s->stream, &grid));
    s->n_part = fuse ? grid : 0;
    return L2Z_OK;
}
}  // namespace

namespace l2z {

// Row-sharded == unsharded bit for bit needs every rank to take the kernel -- the summation order -- the unsharded
// pass takes.  The one kernel with a per-rank shape condition is the K-range panel kernel (a wave's 16 rows lie in one
// matrix: every matrix of a launch must have a multiple of 16 rows ON THE RANK): where some chunk length sends one of
// the WHOLE model's products there and this group's share of its rows is not a multiple of 16, the group keeps off the
// batched path (its prompts are stepped).  A function of the model and the group size: the same on every rank.
bool prefill_shard_takes_the_unsharded_kernels(const l2z_config &c, const Shard &sh)
{
    if (sh.world == 1 || sh.scheme_b) return true;  // (scheme B never takes the panel kernel)
    const long long kvd = (long long)c.dim / c.n_heads * c.n_kv_heads;
    const long long widest_whole = std::max((long long)c.dim + 2 * kvd, 2LL * c.hidden_dim) + 128;
    struct { long long n_whole; int K; bool rows_ok; } prod[4] = {
        {(long long)c.dim + 2 * kvd, c.dim, sh.dim_loc % 16 == 0 && sh.kvd_loc % 16 == 0},
        {c.dim, c.dim, sh.dim_loc % 16 == 0},
        {2LL * c.hidden_dim, c.dim, (2 * sh.hid_loc) % 16 == 0},
        {c.dim, c.hidden_dim, sh.dim_loc % 16 == 0}};
    for (const auto &p : prod) {
        if (p.rows_ok) continue;
        for (int P = 1; P <= prefill_panel_max_tokens(); P++)
            if (prefill_panel_shape(p.n_whole, P, p.K, widest_whole)) return false;
    }
    return true;
}

bool prefill_enabled()
{
    return tunables().prefill != 0;
}

// whether this runstate can take the batched path at all (shape; sharded: a transport that carries
// [chunk, hidden_dim] matrices).  Same answer on every rank of a group.
bool prefill_usable(const l2z_runstate *s)
{
    const l2z_config &c = s->cfg;
    if (c.dim % 4 != 0 || c.hidden_dim % 4 != 0 || s->sh.hs % 4 != 0 || s->sh.hs > 256) return false;
    if (s->sh.scheme_b) {
        // column-sharded Wo / W2: two bulk all-reduces of [chunk, dim] per layer (a 1-rank RCCL group included)
        if (c.dim % s->sh.world != 0 || (c.dim / s->sh.world) % 4 != 0 || s->sh.dim_loc % 4 != 0 || s->sh.hid_loc % 4 != 0) return false;
        if (s->comm == nullptr) return false;
        if (s->comm->world == 1) return s->comm->nccl != nullptr;
        return comm_bulk_ok(s->comm, (size_t)kPrefillChunk * (size_t)c.dim);
    }
    if (s->sh.world == 1) return true;
    if (s->sh.dim_loc % 4 != 0 || s->sh.hid_loc % 4 != 0) return false;
    if (!prefill_shard_takes_the_unsharded_kernels(c, s->sh)) return false;
    const size_t widest = (size_t)(c.dim > c.hidden_dim ? c.dim : c.hidden_dim);
    return comm_bulk_ok(s->comm, (size_t)kPrefillChunk * widest);
}

int prefill_check(const l2z_config *config, const l2z_runstate *s)
{
    L2Z_CHECK(config->dim % 4 == 0 && config->hidden_dim % 4 == 0 && s->sh.hs % 4 == 0 &&
                  s->sh.hs <= 256, L2Z_ERR_INVALID,
              "l2z_prefill: needs dim, hidden_dim, head_size multiples of 4 and head_size <= 256");
    L2Z_CHECK(prefill_usable(s), L2Z_ERR_INVALID,
              "l2z_prefill: this sharded runstate has no transport for [%d, hidden_dim] matrices (RCCL "
              "communicator, or peer-write arena with bulk regions: L2Z_P2P_BULK_MB), or its row shards are "
              "not multiples of 4", kPrefillChunk);
    return L2Z_OK;
}

int prefill_next_chunk_of(const l2z_config &c, int remaining)
{
    const int P = prefill_next_chunk(remaining);
    // 513 ... 1023 tokens on the bf16 cores: 512 first only where the rest is cheap (<= 160 tokens: 576 / 640 tokens 58.0 / 61.2 ms
    // whole, 47.5 / 51.7 cut); beyond, ONE chunk (800 / 832 / 900 / 960 tokens 72.6 / 72.4 / 80.6 / 77.3 cut, 65.8 / 66.0 / 74.4 /
    // 74.9 whole; 704 / 768 tokens equal within 2 %: profiles/r06z_chunk_plan.txt)
    if (tunables().pf_chunk == 0 && remaining > 672 && remaining < 1024 && x3_stream_shape(c.dim, 96, (c.dim + 63) / 64 * 64)) return remaining;
    if (tunables().pf_chunk > 0 || P != remaining || remaining <= prefill_panel_max_tokens()) return P;
    const long long kvd = (long long)c.dim / c.n_heads * c.n_kv_heads;
    const long long widest_whole = std::max((long long)c.dim + 2 * kvd, 2LL * c.hidden_dim) + 128;
    // Wo stands for the model's matrices: a piece "takes the panel kernel" if Wo does at that length (streams from HBM, K % 128 == 0)
    auto panel = [&](int n) { return prefill_panel_shape(c.dim, n, c.dim, widest_whole); };
    if (remaining <= 96) {
        const int first = remaining <= 80 ? 48 : 64;  // 65 ... 80: 48 + 17 ... 32; 81 ... 96: 64 + 17 ... 32
        return panel(first) && panel(remaining - first) ? first : P;
    }
    // Past a step of the tile GEMM's cost staircase the step's worth goes first and the rest -- at most 96 tokens -- to the
    // short-chunk kernels.  On the bf16 cores (round 6; 7B, ms, one chunk against the cut: profiles/r06z_chunk_plan.txt) the
    // tile forms cost 20.9 ... 24.6 for ANY chunk of 129 ... 256 tokens and ~35 for 257 ... 384, the stream form 12.5 at 128 and
    // 5.6 ... 10.5 for a rest of 16 ... 96: 176 / 192 / 224 tokens 23.7 / 24.0 / 24.6 -> 20.2 / 20.3 / 22.8 as 128 + rest (240:
    // 23.1 whole against 25.0), 272 / 320 tokens 34.9 / 35.9 -> 30.2 / 31.1 as 256 + rest.  On the f32 cores the steps were
    // worth 32 tokens only (128 tokens 17.8 ms, 129 ... 160 25.4; 256 tokens 31.6, 288 40.2).
    // (353 ... 384 tokens: 44.2 whole against 35.9 as 256 + 97 ... 128; 225 ... 256 whole: 23.0 against 25.4 as 128 + 97 ... 128)
    const bool stream = x3_stream_shape(c.dim, 96, (c.dim + 63) / 64 * 64);
    for (int q : {128, 256}) {
        const int rest_max = !stream ? 32 : q == 128 ? 96 : 128;
        if (remaining > q && remaining <= q + rest_max) return panel(32) ? q : P;
    }
    return P;
}

// positions pos0 .. pos0+n-1 in chunks; leaves the last position's residual row in RunState.x
int prefill_tokens(l2z_runstate *s, const l2z_weights *w, const int32_t *tokens, int n_tokens,
                          int pos0)
{
    const l2z_config *config = &s->cfg;
    int done = 0;
    while (done < n_tokens) {
        const int P = prefill_next_chunk_of(*config, n_tokens - done);
        L2Z_TRY(prefill_alloc(s, P));
        L2Z_TRY(prefill_chunk(s, w, tokens + done, P, pos0 + done));
        if (done + P == n_tokens)
            L2Z_HIP(hipMemcpyAsync(s->x, s->pf_x + (size_t)(P - 1) * config->dim, (size_t)config->dim * 4,
                                   hipMemcpyDeviceToDevice, s->stream));
        L2Z_HIP(hipStreamSynchronize(s->stream));  // the host token buffer may now be reused
        L2Z_TRY(comm_check(s->comm));
        done += P;
    }
    return L2Z_OK;
}

}  // namespace l2z

extern "C" int l2z_prefill(const int32_t *tokens, int n_tokens, int pos0, const l2z_config *config,
                           l2z_runstate *s, const l2z_weights *w)
{
    L2Z_TRY(check_pair(config, s, w));
    L2Z_CHECK(tokens != nullptr && n_tokens >= 1, L2Z_ERR_INVALID, "l2z_prefill: no tokens");
    L2Z_CHECK(pos0 >= 0 && pos0 + n_tokens <= config->seq_len, L2Z_ERR_STATE,
              "l2z_prefill: positions %d..%d outside [0,%d)", pos0, pos0 + n_tokens - 1, config->seq_len);
    for (int i = 0; i < n_tokens; i++)
        L2Z_CHECK(tokens[i] >= 0 && tokens[i] < config->vocab_size, L2Z_ERR_STATE,
                  "l2z_prefill: tokens[%d] = %d out of vocabulary", i, tokens[i]);
    L2Z_TRY(prefill_check(config, s));
    L2Z_HIP(hipSetDevice(s->device));
    L2Z_TRY(prefill_tokens(s, w, tokens, n_tokens, pos0));
    // the last position's residual row is RunState.x: the usual final rmsnorm + classifier
    // launch (:426-429) leaves the logits in place
    const int last_pos = pos0 + n_tokens - 1;
    L2Z_HIP(hipMemcpyAsync(s->d_pos, &last_pos, sizeof(int), hipMemcpyHostToDevice, s->stream));
    L2Z_HIP(hipMemcpyAsync(s->d_token, &tokens[n_tokens - 1], sizeof(int), hipMemcpyHostToDevice, s->stream));
    L2Z_TRY(prefill_classifier(s, w));
    // sharded: the logits gather, as a pass of its own (it closes the pass: forward.cpp gather())
    if (s->sh.world > 1)
        L2Z_TRY(comm_allgather_inplace(s->comm, s->logits, (size_t)s->sh.v_loc, s->n_gathers, s->n_gathers,
                                       false, s->stream));
    s->logits_partial = false;
    L2Z_HIP(hipStreamSynchronize(s->stream));
    L2Z_TRY(comm_check(s->comm));
    s->host_pos = pos0 + n_tokens;
    return L2Z_OK;
}


// Testing support (include/llama2_hip_test.h): l2z_prefill for N emulated ranks in one process on one
// GPU -- every rank's own launches of each stage, the exchange of the [P, n_loc] blocks as
// device-to-device copies between the ranks' staging buffers, then every rank's own unpack launch.
extern "C" int l2z_emu_prefill(int n_ranks, l2z_runstate *const *ss, const l2z_weights *const *ws,
                               const int32_t *tokens, int n_tokens, int pos0)
{
    L2Z_CHECK(n_ranks >= 2 && ss && ws && tokens && n_tokens >= 1, L2Z_ERR_INVALID, "l2z_emu_prefill: bad arguments");
    const l2z_config &c = ss[0]->cfg;
    L2Z_CHECK(pos0 >= 0 && pos0 + n_tokens <= c.seq_len, L2Z_ERR_STATE, "l2z_emu_prefill: positions out of range");
    for (int i = 0; i < n_tokens; i++)
        L2Z_CHECK(tokens[i] >= 0 && tokens[i] < c.vocab_size, L2Z_ERR_STATE, "l2z_emu_prefill: token out of vocabulary");
    for (int r = 0; r < n_ranks; r++) {
        L2Z_TRY(check_pair(&c, ss[r], ws[r]));
        L2Z_CHECK(ss[r]->sh.world == n_ranks && ss[r]->sh.rank == r, L2Z_ERR_INVALID,
                  "l2z_emu_prefill: runstate %d is not rank %d of %d", r, r, n_ranks);
        L2Z_CHECK(c.dim % 4 == 0 && c.hidden_dim % 4 == 0 && ss[r]->sh.hs % 4 == 0 && ss[r]->sh.hs <= 256 &&
                      ss[r]->sh.dim_loc % 4 == 0 && ss[r]->sh.hid_loc % 4 == 0,
                  L2Z_ERR_INVALID, "l2z_emu_prefill: shape not supported by the batched path");
        L2Z_CHECK(ss[r]->sh.scheme_b == ss[0]->sh.scheme_b, L2Z_ERR_INVALID, "l2z_emu_prefill: the ranks' sharding schemes differ");
        L2Z_CHECK(prefill_shard_takes_the_unsharded_kernels(c, ss[r]->sh), L2Z_ERR_INVALID,
                  "l2z_emu_prefill: a rank's rows are not multiples of 16 where the unsharded pass takes the panel kernel");
        L2Z_TRY(prefill_alloc(ss[r], n_tokens < kPrefillChunk ? n_tokens : kPrefillChunk));
    }
    auto sync_all = [&]() -> int {
        for (int r = 0; r < n_ranks; r++) L2Z_HIP(hipStreamSynchronize(ss[r]->stream));
        return L2Z_OK;
    };
    // rank src's block of `count` floats at `off(rank)` -> every other rank
    auto exchange = [&](size_t count, float *(*buf)(l2z_runstate *)) -> int {
        for (int src = 0; src < n_ranks; src++)
            for (int dst = 0; dst < n_ranks; dst++)
                if (dst != src)
                    L2Z_HIP(hipMemcpy(buf(ss[dst]) + (size_t)src * count, buf(ss[src]) + (size_t)src * count,
                                      count * sizeof(float), hipMemcpyDeviceToDevice));
        L2Z_HIP(hipDeviceSynchronize());  // the ranks' streams are non-blocking (forward.cpp l2z_emu_transformer)
        return L2Z_OK;
    };
    int done = 0;
    while (done < n_tokens) {
        const int P = prefill_next_chunk_of(c, n_tokens - done);
        for (int r = 0; r < n_ranks; r++) L2Z_TRY(prefill_begin_chunk(ss[r], ws[r], tokens + done, P));
        if (ss[0]->sh.scheme_b) {
            // scheme B: every rank's half layer, then the all-reduce -- every rank's pf_x = the partials summed in rank order
            L2Z_CHECK(n_ranks <= kMaxWorld, L2Z_ERR_INVALID, "l2z_emu_prefill: more than %d ranks", kMaxWorld);
            for (int l = 0; l < c.n_layers; l++)
                for (int half = 0; half < 2; half++) {
                    const float *parts[kMaxWorld];
                    for (int r = 0; r < n_ranks; r++) {
                        L2Z_TRY(prefill_half_b(ss[r], ws[r], l, half, P, pos0 + done));
                        L2Z_HIP(hipStreamSynchronize(ss[r]->stream));
                        parts[r] = ss[r]->pf_part;
                    }
                    for (int r = 0; r < n_ranks; r++) L2Z_HIP(launch_sum_parts(ss[r]->pf_x, parts, n_ranks, P * c.dim, nullptr));
                    L2Z_HIP(hipDeviceSynchronize());
                }
        } else
        for (int l = 0; l < c.n_layers; l++)
            for (int k = 0; k < PF_STAGES; k++) {
                // one rank at a time: on hardware every rank has a GPU to itself, so kernel times taken from
                // this driver (scripts/sharded_prefill_emu.py under rocprofv3) should not overlap either
                for (int r = 0; r < n_ranks; r++) {
                    L2Z_TRY(prefill_stage(ss[r], ws[r], l, k, P, pos0 + done));
                    L2Z_HIP(hipStreamSynchronize(ss[r]->stream));
                }
                const StageOut o0 = stage_out(ss[0], k);
                L2Z_TRY(exchange((size_t)P * o0.n_loc, [](l2z_runstate *s) { return s->pf_stage; }));
                for (int r = 0; r < n_ranks; r++) {
                    const StageOut o = stage_out(ss[r], k);
                    BulkArgs a = {};
                    a.stage = ss[r]->pf_stage; a.P = P; a.n_loc = o.n_loc; a.rank = r; a.world = n_ranks;
                    L2Z_HIP(launch_bulk_unpack(a, 0, 0, o.dst, o.ldd, ss[r]->stream));
                    L2Z_HIP(hipStreamSynchronize(ss[r]->stream));
                }
            }
        if (done + P == n_tokens)
            for (int r = 0; r < n_ranks; r++)
                L2Z_HIP(hipMemcpyAsync(ss[r]->x, ss[r]->pf_x + (size_t)(P - 1) * c.dim, (size_t)c.dim * 4,
                                       hipMemcpyDeviceToDevice, ss[r]->stream));
        L2Z_TRY(sync_all());
        done += P;
    }
    const int last_pos = pos0 + n_tokens - 1;
    for (int r = 0; r < n_ranks; r++) {
        L2Z_HIP(hipMemcpyAsync(ss[r]->d_pos, &last_pos, sizeof(int), hipMemcpyHostToDevice, ss[r]->stream));
        L2Z_HIP(hipMemcpyAsync(ss[r]->d_token, &tokens[n_tokens - 1], sizeof(int), hipMemcpyHostToDevice,
                               ss[r]->stream));
        L2Z_TRY(prefill_classifier(ss[r], ws[r]));
    }
    L2Z_TRY(sync_all());
    L2Z_TRY(exchange((size_t)ss[0]->sh.v_loc, [](l2z_runstate *s) { return s->logits; }));
    for (int r = 0; r < n_ranks; r++) ss[r]->host_pos = pos0 + n_tokens;
    return L2Z_OK;
}
