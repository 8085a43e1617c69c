// runstate.cpp -- the device resident RunState (main.zig:119-162) of the C ABI: allocation, the
// host-built RoPE table, reads for tests, the streaming-read probe.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "l2z_state.h"

using namespace l2z;

// ---------------------------------------------------------------------------
// src/main.zig:137 RunState.init
extern "C" int l2z_runstate_init(const l2z_config *config, const l2z_comm *comm, l2z_runstate **out)
{
    L2Z_CHECK(config != nullptr && out != nullptr, L2Z_ERR_INVALID, "runstate_init: null argument");
    const int dev = current_device_for(comm);
    L2Z_TRY(ensure_device(dev));
    Shard sh;
    L2Z_TRY(make_shard(*config, comm, &sh));
    const l2z_config &c = *config;
    const size_t att_lds = attention_lds_bytes(sh.hs, c.seq_len, sh.hs % 4 == 0 && sh.kvd_loc % 4 == 0);
    L2Z_CHECK(att_lds <= 160 * 1024, L2Z_ERR_INVALID,
              "seq_len %d needs %zu bytes of LDS for attention scores (max 163840)", c.seq_len,
              att_lds);
    const int n_max = c.dim > c.hidden_dim ? c.dim : c.hidden_dim;
    L2Z_CHECK(matvec_lds_bytes(n_max) <= 160 * 1024, L2Z_ERR_INVALID,
              "dim/hidden_dim %d does not fit the 160 KiB LDS x-staging buffer", n_max);

    l2z_runstate *s = new l2z_runstate();
    s->cfg = c;
    s->device = dev;
    s->sh = sh;
    s->comm = comm;
    const Tunables &tn = tunables();
    s->max_blocks = tn.max_blocks_per_cu;  // per CU; the launcher also caps at the occupancy query
    s->use_graphs = tn.no_graph == 0;
    // Peer-write gathers are plain kernels (or no launch at all): captured with the rest of the
    // step.  RCCL collectives are captured too (stream capture of ncclAllGather; if the capture
    // fails the step is launched eagerly, ensure_graph).
    // Emulated ranks are driven stage by stage, never captured.
    if (comm && comm->world > 1 && !comm->nccl && !comm->p2p) s->use_graphs = false;
    s->fused_qkv_attn = sh.world == 1 && tn.fuse_small != 0 &&
                        fused_qkv_attn_supported(c.dim, c.n_heads, c.n_kv_heads, c.seq_len, g_cus);
    s->n_gathers = (sh.scheme_b ? 2 : 4) * c.n_layers + 1;
    if (comm_uses_p2p(comm)) {
        // the producers store straight into the peers' landing slots: they must hold the longest vector
        // (scheme B: every rank's whole partial [dim] vector lands in every slot)
        const size_t longest = std::max((size_t)std::max(std::max(c.dim, c.hidden_dim), c.vocab_size),
                                        sh.scheme_b ? (size_t)sh.world * (size_t)c.dim : (size_t)0);
        if (comm->slot_floats < longest) {
            set_error("peer-write landing slots hold %zu floats, this config gathers up to %zu: pass "
                      "max(dim, hidden_dim, vocab_size%s) to l2z_comm_p2p_export", comm->slot_floats, longest,
                      sh.scheme_b ? ", world * dim" : "");
            delete s;
            return L2Z_ERR_COMM;
        }
    }

    const size_t kv = (size_t)c.n_layers * c.seq_len * sh.kvd_loc;
    hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    auto alloc = [&](void **p, size_t bytes) {
        if (e == hipSuccess) e = hipMalloc(p, bytes ? bytes : 4);
        if (e == hipSuccess) e = hipMemset(*p, 0, bytes ? bytes : 4);
    };
    alloc((void **)&s->x, (size_t)c.dim * 4);
    // (+ 256: scheme B's column-shard mat-vecs read their local slice padded to the shard's row width; the pad stays zero)
    alloc((void **)&s->xb, ((size_t)c.dim + 256) * 4);
    alloc((void **)&s->hb, ((size_t)c.hidden_dim + 256) * 4);
    if (sh.scheme_b) alloc((void **)&s->part, (size_t)c.dim * 4);
    alloc((void **)&s->q, (size_t)c.dim * 4);
    alloc((void **)&s->logits, (size_t)c.vocab_size * 4);
    alloc((void **)&s->key_cache, kv * 4);
    alloc((void **)&s->value_cache, kv * 4);
    alloc((void **)&s->rope, (size_t)c.seq_len * (sh.hs / 2) * sizeof(float2));
    alloc((void **)&s->d_token, 4);
    alloc((void **)&s->d_pos, 4);
    alloc((void **)&s->d_n_prompt, 4);
    alloc((void **)&s->d_argmax, 4);
    alloc((void **)&s->d_prompt, (size_t)c.seq_len * 4);
    alloc((void **)&s->d_out_tokens, (size_t)c.seq_len * 4);
    alloc((void **)&s->d_part_val, (size_t)matvec_max_grid(g_cus) * 4);
    alloc((void **)&s->d_part_idx, (size_t)matvec_max_grid(g_cus) * 4);
    {   // Attention form by position.  One block per head is fastest while the
        // context is short; from pos 256 on, the split form (nch blocks per head + combine)
        // wins and keeps winning (2.4x at pos 2047 on the 7B shape); below it the one-block form runs
        // with 256 threads and a speculative first round while the context is short (attn_short_pos),
        // with 1024 threads beyond.  The host knows pos, so it replays the graph captured for the
        // position's variant (forward.cpp attn_variant).  Tunables attn_split: 0 = never, n = n chunks
        // at every position (tests); attn_split_pos / attn_short_pos move the switch-overs.
        const int mode = tn.attn_split;
        // The chunk count is part of the arithmetic (the combine rounds per chunk), so it is taken
        // from the model's TOTAL head count, not this rank's share: sharded and unsharded runs then
        // use the same chunks and stay bit-identical beyond pos 256 as well.
        int nch = mode > 0 ? mode : attention_split_chunks(c.n_heads, g_cus);
        if (nch > 16) nch = 16;
        s->attn_split_pos = mode > 0 ? 0 : 256;
        if (tn.attn_split_pos >= 0) s->attn_split_pos = tn.attn_split_pos;
        if (mode == 0 || c.seq_len <= s->attn_split_pos) nch = 0;
        s->attn_split_wide_pos = attention_split_wide_pos(c.seq_len);
        s->attn_short_pos = attention_short_pos(sh.hs, c.seq_len);
        if (nch > 1 && s->attn_short_pos > s->attn_split_pos) s->attn_short_pos = s->attn_split_pos;
        if (nch > 1) {
            s->attn_nch = nch;
            alloc((void **)&s->d_attn_part, attention_split_part_floats(sh.heads_loc, sh.hs, nch) * 4);
            alloc((void **)&s->d_attn_cnt, (size_t)sh.heads_loc * 4);
        }
    }
    if (comm_uses_p2p(comm) && e == hipSuccess) {
        // consumer-side gathers need every producer to push and every consumer to read LL words:
        // the vector mat-vec kernels and the vector attention kernels
        AttnArgs aa = {};
        aa.q = s->q; aa.kcache = s->key_cache; aa.vcache = s->value_cache;
        aa.head_size = sh.hs; aa.kv_row = sh.hs; aa.kv_head = (size_t)c.seq_len * sh.hs;
        const bool want_consume = tn.p2p_consume >= 0 ? tn.p2p_consume != 0 : (sh.world <= 2 || c.dim < 4096);
        s->ll_consume = !sh.scheme_b && want_consume && matvec_ll_supported(c.dim) &&
                        matvec_ll_supported(c.hidden_dim) && attention_push_supported(aa);
        P2pArgs t[4];
        comm_p2p_args(comm, s->xb, (size_t)sh.dim_loc, s->ll_consume, &t[0]);
        // (scheme B: the pushed vector is the rank's whole partial, element i of rank r at word r * dim + i)
        comm_p2p_args(comm, sh.scheme_b ? s->part : s->x, sh.scheme_b ? (size_t)c.dim : (size_t)sh.dim_loc, s->ll_consume, &t[1]);
        comm_p2p_args(comm, s->hb, (size_t)sh.hid_loc, s->ll_consume, &t[2]);
        comm_p2p_args(comm, s->logits, (size_t)sh.v_loc, false, &t[3]);
        alloc((void **)&s->d_push, sizeof t);
        if (e == hipSuccess) e = hipMemcpy(s->d_push, t, sizeof t, hipMemcpyHostToDevice);
        // greedy steps end in the candidate exchange where the classifier takes the vector kernels (their fused
        // argmax epilogue) and the slots hold the 2 * world words
        s->xchg_steps = tn.argmax_xchg != 0 && matvec_vector_width(c.dim) && sh.v_loc >= 2 && comm->slot_floats >= 2 * (size_t)sh.world;
    }
    if (e != hipSuccess) {
        set_error("RunState allocation failed: %s", hipGetErrorString(e));
        l2z_runstate_free(s);
        return e == hipErrorOutOfMemory ? L2Z_ERR_OOM : L2Z_ERR_HIP;
    }
    // RoPE table: exactly main.zig:338-342 evaluated once per (pos, pair) on the host in f32
    // -- freq = 1/pow(10000, (i % hs)/hs); val = pos*freq; cos(val), sin(val) -- instead of
    // per layer per token on the device (same values for every layer: L-fold less
    // transcendental work, and the same libm the CPU path uses).
    {
        const int half = sh.hs / 2;
        std::vector<float2> tab((size_t)c.seq_len * half);
        for (int j = 0; j < half; j++) {
            const float head_dim = (float)(2 * j);
            const float freq = 1.0f / powf(10000.0f, head_dim / (float)sh.hs);
            for (int p = 0; p < c.seq_len; p++) {
                const float val = (float)p * freq;
                tab[(size_t)p * half + j] = make_float2(cosf(val), sinf(val));
            }
        }
        e = hipMemcpy(s->rope, tab.data(), tab.size() * sizeof(float2), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            set_error("RoPE table upload failed: %s", hipGetErrorString(e));
            l2z_runstate_free(s);
            return L2Z_ERR_HIP;
        }
    }
    *out = s;
    return L2Z_OK;
}

extern "C" void l2z_runstate_free(l2z_runstate *s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (int v = 0; v < l2z::ATTN_VARIANTS; v++) {
        if (s->g_forward[v]) (void)hipGraphExecDestroy(s->g_forward[v]);
        if (s->g_step[v]) (void)hipGraphExecDestroy(s->g_step[v]);
    }
    void *ptrs[] = {s->x, s->xb, s->hb, s->q, s->logits, s->key_cache, s->value_cache, s->rope,
                    s->d_token, s->d_pos, s->d_prompt, s->d_n_prompt, s->d_out_tokens, s->d_argmax,
                    s->d_probs, s->d_part_val, s->d_part_idx, s->d_attn_part, s->d_attn_cnt, s->pf_x, s->pf_xn, s->pf_q,
                    s->pf_att, s->pf_h1, s->pf_stage, s->pf_part, s->pf_tokens, s->d_push, s->pf_sk.part, s->pf_sk.cnt, s->pf_sk.x3, s->pf_sk.x3b, s->part};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (s->h_stage) (void)hipHostFree(s->h_stage);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

// Which structure this runstate runs (a test that asks for an option checks here that it got it): bit 3 sharding
// scheme B (L2Z_SCHEME_B: column-sharded Wo / W2, all-reduces).  Bits 0-2 named round 4's opt-in decode forms
// (paired blocks, two chains, persistent launches), which were measured slower and removed: always 0.
extern "C" int l2z_runstate_form(const l2z_runstate *s, int *form)
{
    L2Z_CHECK(s != nullptr && form != nullptr, L2Z_ERR_INVALID, "l2z_runstate_form: bad arguments");
    *form = s->sh.scheme_b ? 8 : 0;
    return L2Z_OK;
}

extern "C" int l2z_runstate_read(l2z_runstate *s, const char *name, size_t offset, size_t count,
                                 float *out)
{
    L2Z_CHECK(s && name && out, L2Z_ERR_INVALID, "l2z_runstate_read: null argument");
    const l2z_config &c = s->cfg;
    const size_t kv = (size_t)c.n_layers * c.seq_len * s->sh.kvd_loc;
    const float *p = nullptr;
    size_t n = 0;
    const std::string k = name;
    if (k == "x") { p = s->x; n = c.dim; }
    else if (k == "xb") { p = s->xb; n = c.dim; }
    else if (k == "hb") { p = s->hb; n = c.hidden_dim; }
    else if (k == "q") { p = s->q; n = c.dim; }
    else if (k == "logits") { p = s->logits; n = c.vocab_size; }
    else if (k == "key_cache") { p = s->key_cache; n = kv; }
    else if (k == "value_cache") { p = s->value_cache; n = kv; }
    L2Z_CHECK(p != nullptr, L2Z_ERR_INVALID, "l2z_runstate_read: unknown buffer '%s'", name);
    L2Z_CHECK(offset + count <= n, L2Z_ERR_INVALID, "l2z_runstate_read: out of range");
    L2Z_HIP(hipSetDevice(s->device));
    if (k == "logits") L2Z_TRY(ensure_logits(s));
    L2Z_HIP(hipStreamSynchronize(s->stream));
    if (k == "key_cache" || k == "value_cache") {
        // offset / count address the REFERENCE's order (layer, pos, kv_dim) (main.zig:354); a layer of the
        // device cache is head-major, [kv head][pos][head_size]: permute on the way out
        const size_t S = (size_t)c.seq_len, kvd = (size_t)s->sh.kvd_loc, hs = (size_t)s->sh.hs, layer = S * kvd;
        std::vector<float> host(layer);
        for (size_t l = offset / layer; l * layer < offset + count; l++) {
            L2Z_HIP(hipMemcpy(host.data(), p + l * layer, layer * sizeof(float), hipMemcpyDeviceToHost));
            const size_t lo = std::max(offset, l * layer), hi = std::min(offset + count, (l + 1) * layer);
            for (size_t i = lo; i < hi; i++) {
                const size_t r = i - l * layer, t = r / kvd, f = r % kvd;
                out[i - offset] = host[(f / hs) * S * hs + t * hs + f % hs];
            }
        }
        return L2Z_OK;
    }
    L2Z_HIP(hipMemcpy(out, p + offset, count * sizeof(float), hipMemcpyDeviceToHost));
    return L2Z_OK;
}

// Measured ceiling for the roofline: stream `slice_bytes`-sized pieces of the resident weight
// blob through a pure read kernel, a different piece every launch (nothing is re-read from the
// on-die caches unless the blob itself is that small), HIP events on the runstate's stream.
extern "C" int l2z_stream_read_probe(l2z_runstate *s, const l2z_weights *w, size_t slice_bytes, int reps,
                                     double *avg_gbps, double *best_gbps)
{
    L2Z_CHECK(s && w && avg_gbps && best_gbps && reps >= 1, L2Z_ERR_INVALID,
              "l2z_stream_read_probe: bad arguments");
    L2Z_HIP(hipSetDevice(s->device));
    const size_t blob_bytes = w->blob_floats * sizeof(float);
    if (slice_bytes == 0 || slice_bytes > blob_bytes) slice_bytes = blob_bytes;
    slice_bytes &= ~(size_t)4095;
    L2Z_CHECK(slice_bytes >= (1u << 20), L2Z_ERR_INVALID, "l2z_stream_read_probe: blob too small");
    const size_t n_slices = blob_bytes / slice_bytes;
    hipEvent_t e0, e1;
    L2Z_HIP(hipEventCreate(&e0));
    L2Z_HIP(hipEventCreate(&e1));
    double tot = 0.0, best = 1e30;
    int rc = L2Z_OK;
    for (int r = 0; r < reps + 2 && rc == L2Z_OK; r++) {  // two untimed warm-ups
        const float *p = w->blob + (size_t)(r % n_slices) * (slice_bytes / sizeof(float));
        hipError_t e = hipEventRecord(e0, s->stream);
        // d_part_val has one float per possible mat-vec block (8 per CU): the probe's scratch
        if (e == hipSuccess) e = launch_stream_read(p, slice_bytes / sizeof(float), s->d_part_val, g_cus, s->stream);
        if (e == hipSuccess) e = hipEventRecord(e1, s->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess) {
            set_error("l2z_stream_read_probe: %s", hipGetErrorString(e));
            rc = L2Z_ERR_HIP;
            break;
        }
        if (r >= 2) {
            tot += ms;
            if (ms < best) best = ms;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != L2Z_OK) return rc;
    *avg_gbps = (double)slice_bytes / (tot / reps * 1e-3) / 1e9;
    *best_gbps = (double)slice_bytes / (best * 1e-3) / 1e9;
    return L2Z_OK;
}

// The other same-box reference point SURVEY.md 8d names: a device-to-device copy (hipMemcpyAsync) of
// `slice_bytes` pieces of the resident weight blob into a scratch allocation of that size, a different
// piece every time.  Rate in GB/s of bytes COPIED (the memory moves twice that: read + write).
extern "C" int l2z_d2d_copy_probe(l2z_runstate *s, const l2z_weights *w, size_t slice_bytes, int reps,
                                  double *avg_gbps, double *best_gbps)
{
    L2Z_CHECK(s && w && avg_gbps && best_gbps && reps >= 1, L2Z_ERR_INVALID, "l2z_d2d_copy_probe: bad arguments");
    L2Z_HIP(hipSetDevice(s->device));
    const size_t blob_bytes = w->blob_floats * sizeof(float);
    if (slice_bytes == 0 || slice_bytes > blob_bytes) slice_bytes = blob_bytes;
    slice_bytes &= ~(size_t)4095;
    L2Z_CHECK(slice_bytes >= (1u << 20), L2Z_ERR_INVALID, "l2z_d2d_copy_probe: blob too small");
    const size_t n_slices = blob_bytes / slice_bytes;
    void *dst = nullptr;
    L2Z_HIP(hipMalloc(&dst, slice_bytes));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    double tot = 0.0, best = 1e30;
    for (int r = 0; r < reps + 2 && e == hipSuccess; r++) {  // two untimed warm-ups
        const float *p = w->blob + (size_t)(r % n_slices) * (slice_bytes / sizeof(float));
        e = hipEventRecord(e0, s->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dst, p, slice_bytes, hipMemcpyDeviceToDevice, s->stream);
        if (e == hipSuccess) e = hipEventRecord(e1, s->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e == hipSuccess && r >= 2) {
            tot += ms;
            if (ms < best) best = ms;
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(dst);
    if (e != hipSuccess) {
        set_error("l2z_d2d_copy_probe: %s", hipGetErrorString(e));
        return L2Z_ERR_HIP;
    }
    *avg_gbps = (double)slice_bytes / (tot / reps * 1e-3) / 1e9;
    *best_gbps = (double)slice_bytes / (best * 1e-3) / 1e9;
    return L2Z_OK;
}

extern "C" int l2z_synchronize(l2z_runstate *s)
{
    L2Z_CHECK(s != nullptr, L2Z_ERR_INVALID, "l2z_synchronize: null runstate");
    L2Z_HIP(hipSetDevice(s->device));
    L2Z_HIP(hipStreamSynchronize(s->stream));
    L2Z_TRY(comm_check(s->comm));
    return L2Z_OK;
}
