// prefill_host.cpp -- host side of the batched prompt prefill (kernels: prefill.hip).
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
static int prefill_chunk_tokens()
{
    int n = tunables().pf_chunk > 0 ? tunables().pf_chunk : 512;
    if (n < 16) n = 16;
    if (n > 2048) n = 2048;
    return n;
}
#define kPrefillChunk prefill_chunk_tokens()

int prefill_alloc(l2z_runstate *s)
{
    if (s->pf_tokens) return L2Z_OK;  // the last one allocated: all of them exist
    const l2z_config &c = s->cfg;
    const size_t P = kPrefillChunk;
    struct { void **p; size_t bytes; } want[] = {
        {(void **)&s->pf_x, P * c.dim * 4},   {(void **)&s->pf_xn, P * c.dim * 4},
        {(void **)&s->pf_q, P * c.dim * 4},   {(void **)&s->pf_att, P * c.dim * 4},
        {(void **)&s->pf_h1, P * c.hidden_dim * 4},
        {(void **)&s->pf_tokens, P * 4}};
    for (auto &b : want) {
        if (*b.p) continue;  // kept from an earlier, partly failed attempt
        hipError_t e = hipMalloc(b.p, b.bytes);
        if (e != hipSuccess) {
            *b.p = nullptr;
            set_error("prefill scratch allocation (%zu bytes) failed: %s", b.bytes, hipGetErrorString(e));
            return e == hipErrorOutOfMemory ? L2Z_ERR_OOM : L2Z_ERR_HIP;
        }
    }
    return L2Z_OK;
}

int prefill_chunk(l2z_runstate *s, const l2z_weights *w, const int32_t *tokens, int P, int pos0)
{
    const l2z_config &c = s->cfg;
    const Shard &sh = s->sh;
    hipStream_t st = s->stream;
    const int dim = c.dim, hid = c.hidden_dim, kvd = sh.kvd_loc, hs = sh.hs;
    L2Z_HIP(hipMemcpyAsync(s->pf_tokens, tokens, (size_t)P * 4, hipMemcpyHostToDevice, st));
    L2Z_HIP(launch_prefill_embed(s->pf_x, w->tok_emb, s->pf_tokens, dim, P, st));  // :295
    for (int l = 0; l < c.n_layers; l++) {
        float *kc = s->key_cache + (size_t)l * c.seq_len * kvd;
        float *vc = s->value_cache + (size_t)l * c.seq_len * kvd;
        L2Z_HIP(launch_prefill_rmsnorm(s->pf_xn, s->pf_x, w->rms_att + (size_t)l * dim, dim, P, st));  // :305
        L2Z_HIP(launch_prefill_gemm(PG_ROPE, s->pf_xn, dim, w->wq + (size_t)l * dim * dim, s->pf_q, dim,
                                    P, dim, dim, pos0, s->rope, hs, st));                   // :308-351
        L2Z_HIP(launch_prefill_gemm(PG_ROPE_CACHE, s->pf_xn, dim, w->wk + (size_t)l * kvd * dim, kc, kvd,
                                    P, kvd, dim, pos0, s->rope, hs, st));                   // :354-357
        L2Z_HIP(launch_prefill_gemm(PG_CACHE, s->pf_xn, dim, w->wv + (size_t)l * kvd * dim, vc, kvd, P,
                                    kvd, dim, pos0, s->rope, hs, st));                      // :358
        L2Z_HIP(launch_prefill_attention(s->pf_q, dim, kc, vc, s->pf_att, dim, pos0, P, c.n_heads, hs,
                                         kvd, c.n_heads / c.n_kv_heads, c.seq_len, st));    // :361-389
        L2Z_HIP(launch_prefill_gemm(PG_RESID, s->pf_att, dim, w->wo + (size_t)l * dim * dim, s->pf_x, dim,
                                    P, dim, dim, pos0, s->rope, hs, st));                   // :392-395
        L2Z_HIP(launch_prefill_rmsnorm(s->pf_xn, s->pf_x, w->rms_ffn + (size_t)l * dim, dim, P, st));  // :398
        // :405-416: W1 and W3 in one launch with silu(a) * b as its epilogue where the tile kernel
        // takes the shape, else two GEMMs, the second one merging into the first one's output
        const hipError_t pe = launch_prefill_gemm_swiglu_pair(s->pf_xn, dim, w->w1 + (size_t)l * hid * dim,
                                                              w->w3 + (size_t)l * hid * dim, s->pf_h1, hid, P,
                                                              hid, dim, st);
        if (pe == hipErrorNotSupported) {
            L2Z_HIP(launch_prefill_gemm(PG_STORE, s->pf_xn, dim, w->w1 + (size_t)l * hid * dim, s->pf_h1, hid,
                                        P, hid, dim, pos0, s->rope, hs, st));                   // :405
            L2Z_HIP(launch_prefill_gemm(PG_SWIGLU, s->pf_xn, dim, w->w3 + (size_t)l * hid * dim, s->pf_h1, hid,
                                        P, hid, dim, pos0, s->rope, hs, st));   // :408 + :411-416 in the epilogue
        } else {
            L2Z_HIP(pe);
        }
        L2Z_HIP(launch_prefill_gemm(PG_RESID, s->pf_h1, hid, w->w2 + (size_t)l * dim * hid, s->pf_x, dim,
                                    P, dim, hid, pos0, s->rope, hs, st));                   // :419-422
    }
    return L2Z_OK;
}
}  // namespace

namespace l2z {

bool prefill_enabled()
{
    return tunables().prefill != 0;
}

int prefill_check(const l2z_config *config, const l2z_runstate *s)
{
    L2Z_CHECK(s->sh.world == 1, L2Z_ERR_INVALID, "l2z_prefill: not available on a sharded runstate");
    L2Z_CHECK(config->dim % 4 == 0 && config->hidden_dim % 4 == 0 && s->sh.hs % 4 == 0 &&
                  s->sh.hs <= 256, L2Z_ERR_INVALID,
              "l2z_prefill: needs dim, hidden_dim, head_size multiples of 4 and head_size <= 256");
    return L2Z_OK;
}

// positions pos0 .. pos0+n-1 in chunks; leaves the last position's residual row in RunState.x
int prefill_tokens(l2z_runstate *s, const l2z_weights *w, const int32_t *tokens, int n_tokens,
                          int pos0)
{
    const l2z_config *config = &s->cfg;
    L2Z_TRY(prefill_alloc(s));
    int done = 0;
    while (done < n_tokens) {
        const int P = n_tokens - done < kPrefillChunk ? n_tokens - done : kPrefillChunk;
        L2Z_TRY(prefill_chunk(s, w, tokens + done, P, pos0 + done));
        if (done + P == n_tokens)
            L2Z_HIP(hipMemcpyAsync(s->x, s->pf_x + (size_t)(P - 1) * config->dim, (size_t)config->dim * 4,
                                   hipMemcpyDeviceToDevice, s->stream));
        L2Z_HIP(hipStreamSynchronize(s->stream));  // the host token buffer may now be reused
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
    {
        const l2z_config &c = s->cfg;
        MatvecArgs a = {};
        a.w0 = w->wcls; a.out0 = s->logits; a.rows0 = c.vocab_size; a.n = c.dim; a.x = s->x;
        a.rms_w = w->rms_final;
        a.part_val = s->d_part_val; a.part_idx = s->d_part_idx; a.row_offset = 0;
        int grid = 0;
        const bool fuse = matvec_vector_width(c.dim);  // as in enqueue_forward
        L2Z_HIP(launch_matvec(a, PRO_RMS, fuse ? EPI_ARGMAX : EPI_STORE, s->max_blocks, g_cus, s->stream,
                              &grid));
        s->n_part = fuse ? grid : 0;
    }
    L2Z_HIP(hipStreamSynchronize(s->stream));
    s->host_pos = pos0 + n_tokens;
    return L2Z_OK;
}

