// forward.cpp -- transformer() (main.zig:285-430) as a chain of fused launches, its hipGraph
// capture, the on-device greedy loop (main.zig:987-1042 at temperature 0), per-kind profiling and
// the emulated-rank driver.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "l2z_state.h"

namespace l2z {

struct Prof {
    std::vector<hipEvent_t> ev;   // pairs
    std::vector<int> kind;
};

#define L2Z_LAUNCH(kind_id, expr)                                                         \
    do {                                                                                  \
        hipEvent_t _a = nullptr, _b = nullptr;                                            \
        if (prof) {                                                                       \
            L2Z_HIP(hipEventCreate(&_a));                                                 \
            L2Z_HIP(hipEventCreate(&_b));                                                 \
            L2Z_HIP(hipEventRecord(_a, st));                                              \
        }                                                                                 \
        L2Z_HIP(expr);                                                                    \
        if (prof) {                                                                       \
            L2Z_HIP(hipEventRecord(_b, st));                                              \
            prof->ev.push_back(_a);                                                       \
            prof->ev.push_back(_b);                                                       \
            prof->kind.push_back(kind_id);                                                \
        }                                                                                 \
    } while (0)

int check_pair(const l2z_config *config, const l2z_runstate *s, const l2z_weights *w)
{
    L2Z_CHECK(config && s && w, L2Z_ERR_INVALID, "null config / runstate / weights");
    L2Z_CHECK(memcmp(config, &s->cfg, sizeof *config) == 0 &&
                  memcmp(config, &w->cfg, sizeof *config) == 0,
              L2Z_ERR_INVALID, "config does not match the one RunState / Weights were built with");
    L2Z_CHECK(s->device == w->device && s->sh.rank == w->sh.rank && s->sh.world == w->sh.world,
              L2Z_ERR_INVALID, "RunState and Weights live on different devices / shards");
    L2Z_CHECK(s->sh.scheme_b == w->sh.scheme_b, L2Z_ERR_INVALID,
              "RunState and Weights were built under different sharding schemes (L2Z_SCHEME_B changed in between)");
    return L2Z_OK;
}

// The forward pass (main.zig:285-430) as 5 launches per layer + classifier
// (+ argmax/hand-over).  Token and pos are read from device memory.
// `only_stage` >= 0 runs just the launches between two gather points (and no collective):
// the single-process multi-rank emulation (l2z_emu_transformer) interleaves the ranks
// stage by stage and performs the gathers itself.  Stages: 4 per layer (after attention,
// wo, ffn13, ffn2), then the classifier, then argmax.
//
// Sharded runs (world > 1) gather xb, x, hb, x per layer and the logits at the end; gather gi
// (1-based) of the pass is vector (gi-1) % 4 of layer (gi-1) / 4.  Three forms (p2p.hip):
//   ll_consume  producers push LL words to every rank, consumers read them from their own landing
//               slot while staging x: no launch per gather, 5 nodes per layer as at world == 1;
//   gather launch after every producer (peer writes without consumer polling, or RCCL).
// Scheme B (sh.scheme_b: L2Z_SCHEME_B=1 at world > 1; SURVEY.md 8e, the "all-reduce" half of north_star): Wo and W2 are
// sharded by COLUMNS.  A rank's attention output and hidden activations stay local -- wo and w2 read just this rank's slice
// against its columns of every row and leave a partial [dim] vector (rank 0 adds the residual, main.zig:395 / :422), and an
// all-reduce sums the partials in rank order into x on every rank.  2 collectives per layer (+ the logits gather) instead
// of 4; stages between collectives: [qkv, attention, wo], [w1|w3, w2] per layer, then the classifier.  The sum over a row
// is split differently than in the unsharded pass: logits agree at the tolerance of the parity tests, not bit for bit
// (ranks agree with each other exactly: same partials, same order).
int enqueue_forward(l2z_runstate *s, const l2z_weights *w, bool with_step, Prof *prof,
                    int only_stage, int variant, int only_kind)
{
    const bool split = variant == ATTN_SPLIT || variant == ATTN_SPLIT_S;
    const l2z_config &c = s->cfg;
    const Shard &sh = s->sh;
    hipStream_t st = s->stream;
    const l2z_comm *lc = s->comm;
    const size_t dim = c.dim, hid = c.hidden_dim;
    const int mb = s->max_blocks;
    int stage = 0;
    auto want = [&]() { return only_stage < 0 || only_stage == stage; };
    // only_kind >= 0: just the launches of one kind, back to back (l2z_time_kind: kernel duration)
    auto kind = [&](int k) { return only_kind < 0 || only_kind == k; };
    const bool p2p = s->d_push != nullptr && only_stage < 0 && sh.world > 1;
    const bool consume = p2p && s->ll_consume;
    // Producers push their outputs as LL words straight into the peers' slots (the values travel while the launch still
    // runs) only where the CONSUMERS read the words (consumer-side form).  Where a launch collects the vector anyway --
    // gather launches, scheme B's reduce launches -- that launch sends too: a store to a peer's arena from a mat-vec's
    // epilogue holds up the wave's loads behind it (they return in order), which costs the launch more than the earlier
    // departure saves (one rank of 8 alone with free hand-overs, profiles/r04_solo_rank.md: scheme B 580 -> 727 tok/s,
    // gather launches 488 -> 649; that form and its knob are gone since round 6).
    const bool can_push = p2p && consume;
    const bool sb = sh.scheme_b;
    // greedy step of a shard group on the peer-write transport: the classifier leaves this rank's argmax candidates and
    // the hand-over launch exchanges one pair per rank instead of the logits gather + 32000-logit scan (main.zig:715-726
    // over the whole vocabulary all the same: larger value, then lower index)
    const bool xchg = with_step && p2p && s->xchg_steps;
    const int n_g = s->n_gathers;
    int gi = 0;           // gathers issued so far in this pass
    bool pushed = false;  // the launch just made pushed its outputs itself
    const int *ctl = lc ? lc->d_ctl : nullptr;
    auto gather = [&](float *buf, size_t count_per_rank) -> int {
        stage++;
        gi++;
        if (only_stage >= 0 || only_kind >= 0) return L2Z_OK;
        const bool was_pushed = pushed;
        pushed = false;
        if (consume) {
            L2Z_CHECK(was_pushed, L2Z_ERR_STATE, "consumer-side gather %d: the producer did not push", gi);
            if (gi < n_g) return L2Z_OK;  // the consumer collects; only the logits get a launch
        }
        hipEvent_t ea = nullptr, eb = nullptr;
        const bool timed = prof != nullptr && s->comm != nullptr && s->comm->world > 1;
        if (timed) {
            L2Z_HIP(hipEventCreate(&ea));
            L2Z_HIP(hipEventCreate(&eb));
            L2Z_HIP(hipEventRecord(ea, st));
        }
        L2Z_TRY(comm_allgather_inplace(s->comm, buf, count_per_rank, gi, n_g, was_pushed, st));
        if (timed) {
            L2Z_HIP(hipEventRecord(eb, st));
            prof->ev.push_back(ea);
            prof->ev.push_back(eb);
            prof->kind.push_back(KIND_GATHER);
        }
        return L2Z_OK;
    };
    // scheme B: x = sum over ranks of their partial vectors (s->part), as its own launch
    auto reduce = [&]() -> int {
        stage++;
        gi++;
        if (only_stage >= 0 || only_kind >= 0) return L2Z_OK;
        const bool was_pushed = pushed;
        pushed = false;
        hipEvent_t ea = nullptr, eb = nullptr;
        if (prof != nullptr) {
            L2Z_HIP(hipEventCreate(&ea));
            L2Z_HIP(hipEventCreate(&eb));
            L2Z_HIP(hipEventRecord(ea, st));
        }
        L2Z_TRY(comm_allreduce(s->comm, s->part, s->x, dim, gi, was_pushed, st));
        if (prof != nullptr) {
            L2Z_HIP(hipEventRecord(eb, st));
            prof->ev.push_back(ea);
            prof->ev.push_back(eb);
            prof->kind.push_back(KIND_GATHER);
        }
        return L2Z_OK;
    };
    // input of a consumer: the vector gathered as number g (consumer-side form), else the plain buffer
    auto x_in = [&](MatvecArgs &a, const float *plain, int g, size_t count_per_rank) {
        a.x = plain;
        if (consume && g >= 1) a.xin = comm_ll_in(s->comm, g, count_per_rank);
    };
    auto push_to = [&](MatvecArgs &a, int which) {
        if (!can_push) return;
        a.push = s->d_push + which;
        a.push_ctl = ctl;
        a.push_gi = comm_gi(lc, gi + 1);
    };
    // a column-shard mat-vec of scheme B: the local slice x_loc (n_pad floats, zero beyond the slice) against this rank's
    // columns of all `dim` rows -> s->part; rank 0 adds the residual
    auto col_shard = [&](MatvecArgs &a, const float *wcols, const float *x_loc, int n_pad, int *epi) {
        a.w0 = wcols;
        a.x = x_loc;
        a.n = n_pad;
        a.rows0 = c.dim;
        a.out0 = s->part;
        a.resid = sh.rank == 0 ? s->x : nullptr;
        *epi = sh.rank == 0 ? EPI_RESID : EPI_STORE;
        push_to(a, 1);
    };
    for (int l = 0; l < c.n_layers; l++) {
        // :354 loff; inside a layer the device cache is head-major, [kv heads][seq_len][head_size]: the rows
        // one head's attention reads are one contiguous run
        float *kc = s->key_cache + (size_t)l * c.seq_len * sh.kvd_loc;
        float *vc = s->value_cache + (size_t)l * c.seq_len * sh.kvd_loc;
        const size_t kvh_stride = (size_t)c.seq_len * sh.hs;
        const bool fused = s->fused_qkv_attn && !split && only_stage < 0;
        if (fused && kind(KIND_QKV)) {    // small models: :305-389 in one launch, one block per head (fused_small.hip)
            FusedQkvAttnArgs a = {};
            a.wq = w->wq + (size_t)l * dim * dim;
            a.wk = w->wk + (size_t)l * dim * dim;
            a.wv = w->wv + (size_t)l * dim * dim;
            a.rms_w = w->rms_att + (size_t)l * dim; a.x = s->x; a.q_out = s->q;
            a.kcache = kc; a.vcache = vc; a.xb = s->xb; a.pos_ptr = s->d_pos; a.rope = s->rope;
            a.n = c.dim; a.head_size = sh.hs; a.seq_len = c.seq_len; a.kv_dim = sh.kvd_loc;
            a.kv_row = sh.hs; a.kv_head = kvh_stride;
            L2Z_LAUNCH(KIND_QKV, launch_fused_qkv_attn(a, c.n_heads, st));
        }
        if (!fused && want() && kind(KIND_QKV)) {   // rmsnorm (:305) + q,k,v (:308-320) + RoPE (:336-351) + KV write (:354-358)
            MatvecArgs a = {};
            a.w0 = w->wq + (size_t)l * sh.dim_loc * dim;
            a.w1 = w->wk + (size_t)l * sh.kvd_loc * dim;
            a.w2 = w->wv + (size_t)l * sh.kvd_loc * dim;
            a.out0 = s->q; a.out1 = kc; a.out2 = vc;
            a.rows0 = sh.dim_loc; a.rows1 = sh.kvd_loc; a.rows2 = sh.kvd_loc;
            a.pos_stride1 = sh.hs; a.pos_stride2 = sh.hs; a.kv_head_stride = kvh_stride;
            a.n = c.dim; a.rms_w = w->rms_att + (size_t)l * dim;
            x_in(a, s->x, gi, sh.dim_loc);  // layer 0: the embedding row, a plain buffer (gi == 0)
            a.pos_ptr = s->d_pos; a.rope = s->rope; a.head_size = sh.hs; a.rope_segs = 2;
            L2Z_LAUNCH(KIND_QKV, launch_matvec(a, PRO_RMS, EPI_ROPE, mb, g_cus, st));
        }
        if (!fused && want() && kind(KIND_ATTN)) {   // attention (:361-389) over the local heads
            AttnArgs a = {};
            a.q = s->q; a.kcache = kc; a.vcache = vc; a.xb = s->xb + sh.dim0;
            a.pos_ptr = s->d_pos; a.head_size = sh.hs; a.kv_row = sh.hs; a.kv_head = kvh_stride;
            a.kv_mul = c.n_heads / c.n_kv_heads; a.seq_len = c.seq_len;
            a.tl_seq = s->tl_attn_seq++;
            if (!sb && can_push && attention_push_supported(a)) {
                a.push = s->d_push + 0;
                a.push_ctl = ctl;
                a.push_gi = comm_gi(lc, gi + 1);
                pushed = true;
            }
            if (split && s->attn_nch > 1 && attention_split_supported(a))
                L2Z_LAUNCH(KIND_ATTN, launch_attention_split(a, sh.heads_loc, s->attn_nch,
                                                             s->d_attn_part, s->d_attn_cnt, st, variant == ATTN_SPLIT_S));
            else
                L2Z_LAUNCH(KIND_ATTN, launch_attention(a, sh.heads_loc, st, variant == ATTN_SHORT ? 1 : 0));
        }
        if (!sb) L2Z_TRY(gather(s->xb, sh.dim_loc));
        if (want() && kind(KIND_WO)) {   // wo (:392) + residual (:395)
            MatvecArgs a = {};
            int epi = EPI_RESID;
            if (sb) {
                col_shard(a, w->wo + (size_t)l * dim * sh.dimc_pad, s->xb + sh.dim0, sh.dimc_pad, &epi);
            } else {
                a.w0 = w->wo + (size_t)l * sh.dim_loc * dim;
                a.out0 = s->x + sh.dim0; a.resid = s->x + sh.dim0;
                a.rows0 = sh.dim_loc; a.n = c.dim;
                x_in(a, s->xb, gi, sh.dim_loc);
                push_to(a, 1);
            }
            L2Z_LAUNCH(KIND_WO, launch_matvec(a, PRO_NONE, epi, mb, g_cus, st, nullptr, &pushed));
        }
        if (sb) L2Z_TRY(reduce());
        else L2Z_TRY(gather(s->x, sh.dim_loc));
        if (want() && kind(KIND_FFN13)) {   // rmsnorm (:398) + w1,w3 (:405-408) + SiLU*mul (:411-416)
            MatvecArgs a = {};
            a.w0 = w->w1 + (size_t)l * sh.hid_loc * 2 * dim;  // W1 | W3 row-interleaved: one linear sweep
            a.w1 = w->w3 + (size_t)l * sh.hid_loc * 2 * dim;  // = a.w0 + dim; the kernels derive it from a.w0
            a.out0 = sb ? s->hb : s->hb + sh.hid0;  // scheme B: the slice stays local, w2's column shard reads it from the buffer's start
            a.rows0 = sh.hid_loc; a.rows1 = sh.hid_loc; a.n = c.dim;
            a.rms_w = w->rms_ffn + (size_t)l * dim;
            x_in(a, s->x, gi, sh.dim_loc);
            if (!sb) push_to(a, 2);
            L2Z_LAUNCH(KIND_FFN13, launch_matvec(a, PRO_RMS, EPI_SWIGLU, mb, g_cus, st, nullptr, &pushed));
        }
        if (!sb) L2Z_TRY(gather(s->hb, sh.hid_loc));
        if (want() && kind(KIND_FFN2)) {   // w2 (:419) + residual (:422)
            MatvecArgs a = {};
            int epi = EPI_RESID;
            if (sb) {
                col_shard(a, w->w2 + (size_t)l * dim * sh.hidc_pad, s->hb, sh.hidc_pad, &epi);
            } else {
                a.w0 = w->w2 + (size_t)l * sh.dim_loc * hid;
                a.out0 = s->x + sh.dim0; a.resid = s->x + sh.dim0;
                a.rows0 = sh.dim_loc; a.n = c.hidden_dim;
                x_in(a, s->hb, gi, sh.hid_loc);
                push_to(a, 1);
            }
            L2Z_LAUNCH(KIND_FFN2, launch_matvec(a, PRO_NONE, epi, mb, g_cus, st, nullptr, &pushed));
        }
        if (sb) L2Z_TRY(reduce());
        else L2Z_TRY(gather(s->x, sh.dim_loc));
    }
    if (want() && kind(KIND_CLS)) {   // final rmsnorm (:426) + classifier (:429)
        MatvecArgs a = {};
        a.w0 = w->wcls; a.out0 = s->logits + sh.v0;
        a.rows0 = sh.v_loc; a.n = c.dim; a.rms_w = w->rms_final;
        x_in(a, s->x, gi, sh.dim_loc);
        a.part_val = s->d_part_val; a.part_idx = s->d_part_idx; a.row_offset = sh.v0;
        // vector path, single GPU or a shard's exchanging step: the launch also leaves one argmax candidate per block
        // (an emulated rank timed kind by kind takes that form too: it is what a real rank's step launches)
        const bool fuse = (sh.world == 1 || xchg || (only_kind >= 0 && with_step && s->comm && !s->comm->nccl && !s->comm->p2p)) &&
                          matvec_vector_width(c.dim);
        int grid = 0;
        if (!xchg) push_to(a, 3);
        L2Z_LAUNCH(KIND_CLS, launch_matvec(a, PRO_RMS, fuse ? EPI_ARGMAX : EPI_STORE, mb, g_cus, st,
                                           &grid, &pushed));
        s->n_part = fuse ? grid : 0;
        // (a captured graph bakes its form in: run_forward restores the count that belongs to the graph it replays)
        if (only_stage < 0 && only_kind < 0) (with_step ? s->n_part_step : s->n_part_fwd) = s->n_part;
    }
    if (xchg) {  // hand-over n_g of the pass is the candidate exchange inside the argmax launch
        stage++;
        gi++;
        pushed = false;
        if (only_kind < 0) s->logits_partial = true;
    } else {
        L2Z_TRY(gather(s->logits, sh.v_loc));
    }
    if (with_step && want() && kind(KIND_ARGMAX)) {
        ArgmaxArgs a = {};
        a.logits = s->logits; a.vocab = c.vocab_size; a.token_ptr = s->d_token;
        if (s->n_part > 0) { a.part_val = s->d_part_val; a.part_idx = s->d_part_idx; a.n_part = s->n_part; }
        a.pos_ptr = s->d_pos; a.prompt = s->d_prompt; a.n_prompt_ptr = s->d_n_prompt;
        a.out_tokens = s->d_out_tokens; a.argmax_out = s->d_argmax; a.tok_emb = w->tok_emb;
        a.x = s->x; a.dim = c.dim; a.advance = 1;
        if (xchg && only_kind < 0) {
            a.xchg = s->d_push + 3;
            a.xchg_gi = comm_gi(lc, n_g);
            a.epoch_ctl = lc->d_ctl;               // the exchange closes the pass: its epochs are used up
            a.epoch_add = lc->solo ? 0 : n_g;      // (a solo rank's hand-overs all carry index 0)
        }
        L2Z_LAUNCH(KIND_ARGMAX, launch_argmax(a, st));
    }
    return L2Z_OK;
}

int attn_variant(const l2z_runstate *s, int pos)
{
    if (s->attn_nch > 1 && pos >= s->attn_split_pos) return pos < s->attn_split_wide_pos ? ATTN_SPLIT_S : ATTN_SPLIT;
    return pos < s->attn_short_pos ? ATTN_SHORT : ATTN_HEAD;
}

int build_graph(l2z_runstate *s, const l2z_weights *w, bool with_step, int variant,
                hipGraphExec_t *out)
{
    hipGraph_t graph = nullptr;
    L2Z_HIP(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
    int rc = enqueue_forward(s, w, with_step, nullptr, -1, variant);
    hipError_t e = hipStreamEndCapture(s->stream, &graph);
    if (rc != L2Z_OK) {
        if (graph) (void)hipGraphDestroy(graph);
        return rc;
    }
    L2Z_HIP(e);
    e = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    L2Z_HIP(e);
    return L2Z_OK;
}

void drop_graphs(l2z_runstate *s)
{
    for (int v = 0; v < ATTN_VARIANTS; v++) {
        if (s->g_forward[v]) { (void)hipGraphExecDestroy(s->g_forward[v]); s->g_forward[v] = nullptr; }
        if (s->g_step[v]) { (void)hipGraphExecDestroy(s->g_step[v]); s->g_step[v] = nullptr; }
    }
    s->graph_w_uid = 0;
}

// the captured graph of one attention variant (with or without the loop hand-over); a model only ever pays for the
// variants its positions take (attn_variant)
int ensure_graph(l2z_runstate *s, const l2z_weights *w, int variant, bool with_step)
{
    if (!s->use_graphs) return L2Z_OK;
    int rc = L2Z_OK;
    if (s->graph_w_uid != w->uid) {
        // first use with these weights: capture every variant some position of this model takes, now -- not in
        // the middle of a generation loop (the positions either side of every switch-over name them all)
        drop_graphs(s);
        s->graph_w_uid = w->uid;
        const int edges[] = {0, s->attn_short_pos, s->attn_split_pos, s->attn_split_wide_pos, s->cfg.seq_len};
        bool reach[ATTN_VARIANTS] = {};
        for (int e : edges)
            for (int p = e - 1; p <= e; p++)
                if (p >= 0 && p < s->cfg.seq_len) reach[attn_variant(s, p)] = true;
        for (int v = 0; v < ATTN_VARIANTS && rc == L2Z_OK; v++) {
            if (!reach[v]) continue;
            rc = build_graph(s, w, false, v, &s->g_forward[v]);
            if (rc == L2Z_OK) rc = build_graph(s, w, true, v, &s->g_step[v]);
        }
    }
    hipGraphExec_t *slot = with_step ? &s->g_step[variant] : &s->g_forward[variant];
    if (rc == L2Z_OK && *slot) return L2Z_OK;
    if (rc != L2Z_OK || build_graph(s, w, with_step, variant, slot) != L2Z_OK) {
        // capture is an optimisation, not a requirement: run the same launches eagerly
        fprintf(stderr, "llama2_hip: hipGraph capture failed (%s); launching eagerly\n", l2z_last_error());
        drop_graphs(s);
        s->use_graphs = false;
        (void)hipGetLastError();
    }
    return L2Z_OK;
}

// one forward pass at position `pos` (the host mirrors the device-side pos)
int run_forward(l2z_runstate *s, const l2z_weights *w, bool with_step, int pos)
{
    const int variant = attn_variant(s, pos);
    L2Z_CHECK(s->sh.world == 1 || (s->comm && (s->comm->nccl || s->comm->p2p)), L2Z_ERR_STATE,
              "sharded runstate without a transport: connect the group (RCCL id or "
              "l2z_comm_p2p_export/_connect), or drive emulated ranks with l2z_emu_transformer");
    L2Z_TRY(comm_check(s->comm));
    L2Z_TRY(ensure_graph(s, w, variant, with_step));
    s->logits_partial = with_step && s->xchg_steps && s->d_push != nullptr && s->sh.world > 1;  // enqueue_forward's `xchg`
    if (s->use_graphs) {
        s->n_part = with_step ? s->n_part_step : s->n_part_fwd;
        L2Z_HIP(hipGraphLaunch(with_step ? s->g_step[variant] : s->g_forward[variant], s->stream));
        return L2Z_OK;
    }
    return enqueue_forward(s, w, with_step, nullptr, -1, variant);
}

// After a greedy step that ended in the candidate exchange `logits` holds this rank's rows only: gather the rest, as a
// pass of one hand-over (collective: every rank of the group calls whatever reads the whole vector, like every other call)
int ensure_logits(l2z_runstate *s)
{
    if (!s->logits_partial) return L2Z_OK;
    L2Z_TRY(comm_allgather_inplace(s->comm, s->logits, (size_t)s->sh.v_loc, 1, 1, false, s->stream));
    s->logits_partial = false;
    s->n_part = 0;  // the candidates named this rank's rows only
    return L2Z_OK;
}


}  // namespace l2z

using namespace l2z;

// src/main.zig:285 transformer(token, pos, config, s, w)
extern "C" int l2z_transformer(int token, int pos, const l2z_config *config, l2z_runstate *s,
                               const l2z_weights *w)
{
    L2Z_TRY(check_pair(config, s, w));
    L2Z_CHECK(token >= 0 && token < config->vocab_size, L2Z_ERR_STATE, "token %d out of range", token);
    L2Z_CHECK(pos >= 0 && pos < config->seq_len, L2Z_ERR_STATE, "pos %d out of range [0,%d)", pos,
              config->seq_len);
    L2Z_HIP(hipSetDevice(s->device));
    L2Z_HIP(launch_set_state(token, pos, s->d_token, s->d_pos, w->tok_emb, s->x, config->dim,
                             s->stream));
    L2Z_TRY(run_forward(s, w, false, pos));
    s->host_pos = pos + 1;
    return L2Z_OK;
}

extern "C" int l2z_argmax(l2z_runstate *s, int *out_token)
{
    L2Z_CHECK(s && out_token, L2Z_ERR_INVALID, "l2z_argmax: null argument");
    L2Z_HIP(hipSetDevice(s->device));
    L2Z_TRY(ensure_logits(s));
    ArgmaxArgs a = {};
    a.logits = s->logits; a.vocab = s->cfg.vocab_size; a.argmax_out = s->d_argmax; a.advance = 0;
    if (s->n_part > 0) { a.part_val = s->d_part_val; a.part_idx = s->d_part_idx; a.n_part = s->n_part; }
    L2Z_HIP(launch_argmax(a, s->stream));
    L2Z_HIP(hipMemcpyAsync(out_token, s->d_argmax, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    L2Z_HIP(hipStreamSynchronize(s->stream));
    return L2Z_OK;
}

extern "C" int l2z_logits_read(l2z_runstate *s, float *out_logits)
{
    L2Z_CHECK(s && out_logits, L2Z_ERR_INVALID, "l2z_logits_read: null argument");
    L2Z_HIP(hipSetDevice(s->device));
    L2Z_TRY(ensure_logits(s));
    L2Z_HIP(hipMemcpyAsync(out_logits, s->logits, (size_t)s->cfg.vocab_size * sizeof(float),
                           hipMemcpyDeviceToHost, s->stream));
    L2Z_HIP(hipStreamSynchronize(s->stream));
    L2Z_TRY(comm_check(s->comm));
    return L2Z_OK;
}

// src/main.zig:1005-1008 on the device: softmax(logits / temperature), then the D2H the samplers need anyway
extern "C" int l2z_probs_read(l2z_runstate *s, float temperature, float *out_probs)
{
    L2Z_CHECK(s && out_probs, L2Z_ERR_INVALID, "l2z_probs_read: null argument");
    L2Z_CHECK(temperature > 0.0f, L2Z_ERR_INVALID, "l2z_probs_read: temperature %g (0 is the argmax path)", (double)temperature);
    L2Z_HIP(hipSetDevice(s->device));
    L2Z_TRY(ensure_logits(s));
    if (s->d_probs == nullptr) L2Z_HIP(hipMalloc(&s->d_probs, (size_t)s->cfg.vocab_size * sizeof(float)));
    L2Z_HIP(launch_probs(s->d_probs, s->logits, s->cfg.vocab_size, temperature, s->stream));
    // through a pinned buffer: a copy into pageable memory is staged by the runtime anyway, slower
    const size_t bytes = (size_t)s->cfg.vocab_size * sizeof(float);
    if (s->h_stage == nullptr) L2Z_HIP(hipHostMalloc((void **)&s->h_stage, bytes, hipHostMallocDefault));
    L2Z_HIP(hipMemcpyAsync(s->h_stage, s->d_probs, bytes, hipMemcpyDeviceToHost, s->stream));
    L2Z_HIP(hipStreamSynchronize(s->stream));
    L2Z_TRY(comm_check(s->comm));
    memcpy(out_probs, s->h_stage, bytes);
    return L2Z_OK;
}

// ---------------------------------------------------------------------------
// src/main.zig:987-1042 at temperature 0
extern "C" int l2z_greedy_begin(l2z_runstate *s, const int32_t *prompt, int n_prompt)
{
    L2Z_CHECK(s != nullptr && n_prompt >= 0 && (n_prompt == 0 || prompt != nullptr),
              L2Z_ERR_INVALID, "l2z_greedy_begin: bad arguments");
    L2Z_CHECK(n_prompt <= s->cfg.seq_len, L2Z_ERR_INVALID, "prompt longer than seq_len");
    for (int i = 0; i < n_prompt; i++)
        L2Z_CHECK(prompt[i] >= 0 && prompt[i] < s->cfg.vocab_size, L2Z_ERR_INVALID,
                  "prompt[%d] = %d out of vocabulary", i, prompt[i]);
    L2Z_HIP(hipSetDevice(s->device));
    L2Z_HIP(hipStreamSynchronize(s->stream));
    if (n_prompt)
        L2Z_HIP(hipMemcpy(s->d_prompt, prompt, (size_t)n_prompt * sizeof(int), hipMemcpyHostToDevice));
    L2Z_HIP(hipMemcpy(s->d_n_prompt, &n_prompt, sizeof(int), hipMemcpyHostToDevice));
    s->h_prompt.assign(prompt, prompt + n_prompt);
    s->host_pos = 0;
    s->done = false;
    return L2Z_OK;
}

extern "C" int l2z_greedy_run(const l2z_config *config, l2z_runstate *s, const l2z_weights *w,
                              int n_steps, int32_t *out_tokens, int *out_n)
{
    L2Z_TRY(check_pair(config, s, w));
    L2Z_CHECK(out_tokens && out_n && n_steps >= 0, L2Z_ERR_INVALID, "l2z_greedy_run: bad arguments");
    *out_n = 0;
    L2Z_HIP(hipSetDevice(s->device));
    if (s->done) return L2Z_OK;
    int remaining = n_steps;
    if (remaining > config->seq_len - s->host_pos) remaining = config->seq_len - s->host_pos;
    if (remaining <= 0) return L2Z_OK;
    if (s->host_pos == 0) {
        // token = 1 (BOS, main.zig:988), pos = 0, x = embedding row of BOS
        L2Z_HIP(launch_set_state(1, 0, s->d_token, s->d_pos, w->tok_emb, s->x, config->dim,
                                 s->stream));
    }
    const int kChunk = 64;  // host looks for BOS (main.zig:1017) once per chunk
    int produced = 0;
    // Prompt positions (main.zig:999-1000 forces next = prompt[pos], the logits there are never
    // looked at) run as one batched pass: inputs BOS, prompt[0..n-2] at positions 0..n-1, then
    // the loop resumes at pos = n with token = prompt[n-1].  Only when this call covers the whole
    // prompt, no prompt token is BOS (the loop would stop there, :1017) and L2Z_PREFILL != 0.
    const int np = (int)s->h_prompt.size();
    if (s->host_pos == 0 && np >= kPrefillMinPrompt && remaining >= np && prefill_enabled() &&
        prefill_usable(s) &&
        std::find(s->h_prompt.begin(), s->h_prompt.end(), 1) == s->h_prompt.end()) {
        std::vector<int32_t> in((size_t)np);
        in[0] = 1;
        for (int i = 1; i < np; i++) in[(size_t)i] = s->h_prompt[(size_t)i - 1];
        L2Z_TRY(prefill_tokens(s, w, in.data(), np, 0));
        L2Z_HIP(hipMemcpyAsync(s->d_out_tokens, s->d_prompt, (size_t)np * sizeof(int),
                               hipMemcpyDeviceToDevice, s->stream));
        L2Z_HIP(launch_set_state(s->h_prompt[(size_t)np - 1], np, s->d_token, s->d_pos, w->tok_emb,
                                 s->x, config->dim, s->stream));
        for (int i = 0; i < np; i++) out_tokens[i] = s->h_prompt[(size_t)i];
        produced = np;
        s->host_pos = np;
        remaining -= np;
    }
    while (remaining > 0 && !s->done) {
        const int n = remaining < kChunk ? remaining : kChunk;
        for (int i = 0; i < n; i++) L2Z_TRY(run_forward(s, w, true, s->host_pos + i));
        L2Z_HIP(hipMemcpyAsync(out_tokens + produced, s->d_out_tokens + s->host_pos,
                               (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s->stream));
        L2Z_HIP(hipStreamSynchronize(s->stream));
        L2Z_TRY(comm_check(s->comm));
            int got = n;
        for (int i = 0; i < n; i++) {
            if (out_tokens[produced + i] == 1) {  // BOS ends the sequence
                got = i + 1;
                s->done = true;
                break;
            }
        }
        produced += got;
        s->host_pos += got;
        remaining -= n;
    }
    *out_n = produced;
    return L2Z_OK;
}

// ---------------------------------------------------------------------------
// One forward pass launched eagerly with a HIP event pair around every kernel:
// per-kind device time, measured live on the stream the kernels run on.
extern "C" int l2z_profile_forward(int token, int pos, const l2z_config *config, l2z_runstate *s,
                                   const l2z_weights *w, double *ms_by_kind, int *launches_by_kind,
                                   int n_kinds)
{
    L2Z_TRY(check_pair(config, s, w));
    L2Z_CHECK(ms_by_kind && launches_by_kind && n_kinds >= KIND_COUNT, L2Z_ERR_INVALID,
              "l2z_profile_forward: need %d kind slots", (int)KIND_COUNT);
    L2Z_CHECK(token >= 0 && token < config->vocab_size && pos >= 0 && pos < config->seq_len,
              L2Z_ERR_STATE, "token/pos out of range");
    L2Z_HIP(hipSetDevice(s->device));
    L2Z_HIP(launch_set_state(token, pos, s->d_token, s->d_pos, w->tok_emb, s->x, config->dim,
                             s->stream));
    Prof prof;
    int rc = enqueue_forward(s, w, true, &prof, -1, attn_variant(s, pos));
    hipError_t e = hipStreamSynchronize(s->stream);
    for (int k = 0; k < n_kinds; k++) { ms_by_kind[k] = 0.0; launches_by_kind[k] = 0; }
    if (rc == L2Z_OK && e == hipSuccess) {
        for (size_t i = 0; i < prof.kind.size(); i++) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, prof.ev[2 * i], prof.ev[2 * i + 1]) == hipSuccess) {
                ms_by_kind[prof.kind[i]] += ms;
                launches_by_kind[prof.kind[i]] += 1;
            }
        }
    }
    for (hipEvent_t ev : prof.ev) (void)hipEventDestroy(ev);
    s->host_pos = pos + 1;
    if (rc != L2Z_OK) return rc;
    L2Z_HIP(e);
    return L2Z_OK;
}

// Duration of ONE kind of launch without the per-launch event pair of l2z_profile_forward (an
// event pair adds ~3 us to a 10-50 us kernel): the launches of that kind for every layer go out back
// to back -- each streams its own layer's weights, nothing is re-read from cache -- between one event
// pair on the runstate's stream; the result is the average per launch, directly comparable with
// rocprofv3's kernel durations.  Unsharded runstates, or emulated ranks of a shard group.
extern "C" int l2z_time_kind(int kind, int pos, const l2z_config *config, l2z_runstate *s,
                             const l2z_weights *w, int reps, double *avg_ms_per_launch, int *launches)
{
    L2Z_TRY(check_pair(config, s, w));
    L2Z_CHECK(kind >= 0 && kind < KIND_GATHER && avg_ms_per_launch && reps >= 1, L2Z_ERR_INVALID,
              "l2z_time_kind: bad arguments");
    // a shard's launches can be timed on an EMULATED rank (no transport: every input is a plain buffer, nothing waits):
    // the per-rank kernel time of a sharded pass, measured on one GPU (bench.py extra.scaling_model)
    L2Z_CHECK(s->sh.world == 1 || (s->comm && ((!s->comm->nccl && !s->comm->p2p) || s->comm->solo)), L2Z_ERR_INVALID,
              "l2z_time_kind: unsharded runstates, emulated or solo ranks only (a connected shard would wait for its peers)");
    L2Z_CHECK(pos >= 0 && pos < config->seq_len, L2Z_ERR_STATE, "pos out of range");
    L2Z_HIP(hipSetDevice(s->device));
    L2Z_HIP(launch_set_state(1, pos, s->d_token, s->d_pos, w->tok_emb, s->x, config->dim, s->stream));
    hipEvent_t e0, e1;
    L2Z_HIP(hipEventCreate(&e0));
    L2Z_HIP(hipEventCreate(&e1));
    const int per_pass = kind == KIND_CLS || kind == KIND_ARGMAX ? 1 : config->n_layers;
    int rc = enqueue_forward(s, w, true, nullptr, -1, attn_variant(s, pos), kind);  // warm-up pass
    hipError_t e = hipEventRecord(e0, s->stream);
    for (int r = 0; r < reps && rc == L2Z_OK; r++) rc = enqueue_forward(s, w, true, nullptr, -1, attn_variant(s, pos), kind);
    if (e == hipSuccess) e = hipEventRecord(e1, s->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.0f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    // the argmax kind advanced pos; put the loop state back
    hipError_t e2 = launch_set_state(1, pos, s->d_token, s->d_pos, w->tok_emb, s->x, config->dim, s->stream);
    if (e2 == hipSuccess) e2 = hipStreamSynchronize(s->stream);
    if (rc != L2Z_OK) return rc;
    L2Z_HIP(e);
    L2Z_HIP(e2);
    *avg_ms_per_launch = (double)ms / (double)(reps * per_pass);
    if (launches) *launches = reps * per_pass;
    return L2Z_OK;
}

// ---------------------------------------------------------------------------
// Single-process emulation of an N-rank shard group on ONE GPU (testing support): the
// dev box has one GPU and RCCL refuses two ranks on one device, so this runs the exact
// per-rank launches of enqueue_forward for every emulated rank, stage by stage, and does
// each all-gather as device-to-device copies.  Validates sharded upload, shard offsets,
// KV-cache sharding and GQA head mapping of the real HIP code without a second GPU.
extern "C" int l2z_emu_transformer(int n_ranks, l2z_runstate *const *ss,
                                   const l2z_weights *const *ws, int token, int pos)
{
    L2Z_CHECK(n_ranks >= 1 && ss && ws, L2Z_ERR_INVALID, "l2z_emu_transformer: bad arguments");
    const l2z_config &c = ss[0]->cfg;
    for (int r = 0; r < n_ranks; r++) {
        L2Z_TRY(check_pair(&c, ss[r], ws[r]));
        L2Z_CHECK(ss[r]->sh.world == n_ranks && ss[r]->sh.rank == r, L2Z_ERR_INVALID,
                  "l2z_emu_transformer: runstate %d is not rank %d of %d", r, r, n_ranks);
        L2Z_HIP(launch_set_state(token, pos, ss[r]->d_token, ss[r]->d_pos, ws[r]->tok_emb, ss[r]->x,
                                 c.dim, ss[r]->stream));
    }
    const bool sb = ss[0]->sh.scheme_b;
    const int n_stages = (sb ? 2 : 4) * c.n_layers + 1;
    for (int stage = 0; stage < n_stages; stage++) {
        for (int r = 0; r < n_ranks; r++)
            L2Z_TRY(enqueue_forward(ss[r], ws[r], false, nullptr, stage, attn_variant(ss[r], pos)));
        for (int r = 0; r < n_ranks; r++) L2Z_HIP(hipStreamSynchronize(ss[r]->stream));
        if (sb && stage < n_stages - 1) {
            // scheme B: the all-reduce -- every rank's x = the partials summed in rank order
            const float *parts[kMaxWorld];
            L2Z_CHECK(n_ranks <= kMaxWorld, L2Z_ERR_INVALID, "l2z_emu_transformer: more than %d ranks", kMaxWorld);
            for (int r = 0; r < n_ranks; r++) parts[r] = ss[r]->part;
            for (int r = 0; r < n_ranks; r++) L2Z_HIP(launch_sum_parts(ss[r]->x, parts, n_ranks, c.dim, nullptr));
            L2Z_HIP(hipDeviceSynchronize());
            continue;
        }
        // which buffer this stage produced, and the per-rank slice length
        const Shard &sh0 = ss[0]->sh;
        size_t count;
        int which;  // 0 xb, 1 x, 2 hb, 3 logits
        if (stage == n_stages - 1) { which = 3; count = sh0.v_loc; }
        else if (stage % 4 == 0) { which = 0; count = sh0.dim_loc; }
        else if (stage % 4 == 2) { which = 2; count = sh0.hid_loc; }
        else { which = 1; count = sh0.dim_loc; }
        auto buf = [&](l2z_runstate *s) {
            return which == 0 ? s->xb : which == 1 ? s->x : which == 2 ? s->hb : s->logits;
        };
        for (int src = 0; src < n_ranks; src++)
            for (int dst = 0; dst < n_ranks; dst++)
                if (dst != src)
                    L2Z_HIP(hipMemcpy(buf(ss[dst]) + (size_t)src * count,
                                      buf(ss[src]) + (size_t)src * count, count * sizeof(float),
                                      hipMemcpyDeviceToDevice));
        // D2D hipMemcpy runs on the null stream and may return early; the ranks' streams are
        // non-blocking, so order the next stage behind the copies explicitly
        L2Z_HIP(hipDeviceSynchronize());
    }
    for (int r = 0; r < n_ranks; r++) ss[r]->host_pos = pos + 1;
    return L2Z_OK;
}

extern "C" int l2z_kind_name(int kind, char *out, size_t cap)
{
    L2Z_CHECK(kind >= 0 && kind < KIND_COUNT && out && cap, L2Z_ERR_INVALID, "l2z_kind_name: bad kind");
    snprintf(out, cap, "%s", kKindNames[kind]);
    return L2Z_OK;
}

