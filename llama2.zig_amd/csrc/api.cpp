// api.cpp -- the C ABI of include/llama2_hip.h: device-resident Weights and
// RunState, the forward pass as a chain of fused launches, its hipGraph
// capture, the on-device greedy loop, and the kernel-level test hooks.
//
// Product code.  No CPU fallback anywhere: without a HIP device every compute
// entry point returns L2Z_ERR_NO_DEVICE.  Nothing under oracle/ is referenced.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "l2z_comm.h"
#include "l2z_internal.h"

namespace l2z {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

namespace {

// ----- shard geometry (DESIGN.md "Sharding"; world == 1 -> everything local) -----
struct Shard {
    int rank = 0, world = 1;
    int hs = 0;        // head_size
    int dim0 = 0, dim_loc = 0;   // rows of q / wo / w2 and slice of x, xb owned here
    int kvd_loc = 0;             // local kv_dim (whole kv heads)
    int heads_loc = 0;
    int hid0 = 0, hid_loc = 0;   // rows of w1/w3, slice of hb
    int v0 = 0, v_loc = 0;       // rows of wcls, slice of logits
};

int make_shard(const l2z_config &c, const l2z_comm *comm, Shard *out)
{
    Shard s;
    s.rank = comm ? comm->rank : 0;
    s.world = comm ? comm->world : 1;
    L2Z_CHECK(c.dim > 0 && c.hidden_dim > 0 && c.n_layers > 0 && c.n_heads > 0 &&
                  c.n_kv_heads > 0 && c.vocab_size > 0 && c.seq_len > 0,
              L2Z_ERR_INVALID, "config: all seven fields must be positive");
    L2Z_CHECK(c.dim % c.n_heads == 0, L2Z_ERR_INVALID, "config: dim %% n_heads != 0");
    L2Z_CHECK(c.n_heads % c.n_kv_heads == 0, L2Z_ERR_INVALID, "config: n_heads %% n_kv_heads != 0");
    s.hs = c.dim / c.n_heads;
    L2Z_CHECK(s.hs % 2 == 0, L2Z_ERR_INVALID, "config: head_size must be even (RoPE pairs)");
    const int kv_dim = (int)(((int64_t)c.dim * c.n_kv_heads) / c.n_heads);
    int64_t a, b;
    L2Z_TRY(l2z_shard_range(c.dim, s.hs, s.rank, s.world, &a, &b));
    s.dim0 = (int)a;
    s.dim_loc = (int)(b - a);
    s.heads_loc = s.dim_loc / s.hs;
    L2Z_TRY(l2z_shard_range(kv_dim, s.hs, s.rank, s.world, &a, &b));
    s.kvd_loc = (int)(b - a);
    L2Z_TRY(l2z_shard_range(c.hidden_dim, 1, s.rank, s.world, &a, &b));
    s.hid0 = (int)a;
    s.hid_loc = (int)(b - a);
    L2Z_TRY(l2z_shard_range(c.vocab_size, 1, s.rank, s.world, &a, &b));
    s.v0 = (int)a;
    s.v_loc = (int)(b - a);
    *out = s;
    return L2Z_OK;
}

// ----- the Weights.init pointer walk (main.zig:85-112) as a table -----
enum ShardKind { REPL, BY_Q_HEADS, BY_KV_HEADS, BY_HIDDEN, BY_DIM_ROWS, BY_VOCAB, SKIP };

struct TensorDesc {
    const char *name;
    size_t offset;  // f32 index in the file blob
    size_t layers, rows, cols;
    ShardKind kind;
    float scale, bias;  // synthetic generator
    size_t count() const { return layers * rows * cols; }
};

std::vector<TensorDesc> tensor_table(const l2z_config &c, bool shared)
{
    const size_t V = c.vocab_size, dim = c.dim, hid = c.hidden_dim, L = c.n_layers;
    const size_t S = c.seq_len, hs = dim / c.n_heads, kvd = (dim * c.n_kv_heads) / c.n_heads;
    const float s_dim = sqrtf(3.0f / (float)dim), s_hid = sqrtf(3.0f / (float)hid);
    const float s_emb = 2.0f * s_dim;
    std::vector<TensorDesc> t = {
        {"token_embedding_table", 0, 1, V, dim, REPL, s_emb, 0.0f},   // :86
        {"rms_att_weight", 0, L, 1, dim, REPL, 0.1f, 1.0f},           // :88
        {"wq", 0, L, dim, dim, BY_Q_HEADS, s_dim, 0.0f},              // :90
        {"wk", 0, L, kvd, dim, BY_KV_HEADS, s_dim, 0.0f},             // :92
        {"wv", 0, L, kvd, dim, BY_KV_HEADS, s_dim, 0.0f},             // :94
        {"wo", 0, L, dim, dim, BY_DIM_ROWS, s_dim, 0.0f},             // :96
        {"rms_ffn_weight", 0, L, 1, dim, REPL, 0.1f, 1.0f},           // :98
        {"w1", 0, L, hid, dim, BY_HIDDEN, s_dim, 0.0f},               // :100
        {"w2", 0, L, dim, hid, BY_DIM_ROWS, s_hid, 0.0f},             // :102
        {"w3", 0, L, hid, dim, BY_HIDDEN, s_dim, 0.0f},               // :104
        {"rms_final_weight", 0, 1, 1, dim, REPL, 0.1f, 1.0f},         // :106
        {"freq_cis_real", 0, 1, 1, S * hs / 2, SKIP, 1.0f, 0.0f},     // :108 (never read)
        {"freq_cis_imag", 0, 1, 1, S * hs / 2, SKIP, 1.0f, 0.0f},     // :110
    };
    if (!shared) t.push_back({"wcls", 0, 1, V, dim, BY_VOCAB, s_emb, 0.0f});  // :112
    size_t off = 0;
    for (auto &d : t) {
        d.offset = off;
        off += d.count();
    }
    return t;
}

void shard_rows(const TensorDesc &d, const Shard &s, size_t *r0, size_t *r1)
{
    switch (d.kind) {
        case BY_Q_HEADS:
        case BY_DIM_ROWS: *r0 = s.dim0; *r1 = (size_t)s.dim0 + s.dim_loc; break;
        case BY_KV_HEADS: *r0 = (size_t)s.rank * s.kvd_loc; *r1 = *r0 + s.kvd_loc; break;
        case BY_HIDDEN: *r0 = s.hid0; *r1 = (size_t)s.hid0 + s.hid_loc; break;
        case BY_VOCAB: *r0 = s.v0; *r1 = (size_t)s.v0 + s.v_loc; break;
        default: *r0 = 0; *r1 = d.rows; break;
    }
}

int g_cus = 0;

int ensure_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available (%s); this library has no CPU fallback",
                  e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return L2Z_ERR_NO_DEVICE;
    }
    L2Z_CHECK(device >= 0 && device < n, L2Z_ERR_NO_DEVICE, "device %d out of range (%d present)",
              device, n);
    L2Z_HIP(hipSetDevice(device));
    if (g_cus == 0) {
        hipDeviceProp_t p;
        L2Z_HIP(hipGetDeviceProperties(&p, device));
        g_cus = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }
    return L2Z_OK;
}

int current_device_for(const l2z_comm *comm)
{
    if (comm) return comm->device;
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) d = 0;
    return d;
}

}  // namespace
}  // namespace l2z

using namespace l2z;

// ---------------------------------------------------------------------------
struct l2z_weights {
    l2z_config cfg;
    int shared;
    int device;
    Shard sh;
    float *blob = nullptr;       // one allocation; world==1: identical to the file blob
    size_t blob_floats = 0;
    bool file_layout = false;
    // carved device pointers (local shards when world > 1)
    const float *tok_emb = nullptr, *rms_att = nullptr, *rms_ffn = nullptr, *rms_final = nullptr;
    const float *wq = nullptr, *wk = nullptr, *wv = nullptr, *wo = nullptr;
    const float *w1 = nullptr, *w2 = nullptr, *w3 = nullptr, *wcls = nullptr;
};

enum { KIND_QKV = 0, KIND_ATTN, KIND_WO, KIND_FFN13, KIND_FFN2, KIND_CLS, KIND_ARGMAX, KIND_COUNT };
static const char *kKindNames[KIND_COUNT] = {"qkv", "attn", "wo", "ffn13", "ffn2", "cls", "argmax"};

struct l2z_runstate {
    l2z_config cfg;
    int device;
    Shard sh;
    const l2z_comm *comm = nullptr;
    hipStream_t stream = nullptr;
    // main.zig:119-135 (k, v, xb2, hb2, logits_indexed have no device twin:
    // k/v go straight into the cache rows, xb2/hb2 are fused away)
    float *x = nullptr, *xb = nullptr, *hb = nullptr, *q = nullptr, *logits = nullptr;
    float *key_cache = nullptr, *value_cache = nullptr;
    float2 *rope = nullptr;  // (seq_len, head_size/2) {cos, sin}
    // loop state on the device
    int *d_token = nullptr, *d_pos = nullptr, *d_prompt = nullptr, *d_n_prompt = nullptr;
    int *d_out_tokens = nullptr, *d_argmax = nullptr;
    // batched prefill scratch (allocated on first l2z_prefill): [kPrefillChunk, dim|hidden]
    float *pf_x = nullptr, *pf_xn = nullptr, *pf_q = nullptr, *pf_att = nullptr;
    float *pf_h1 = nullptr;
    int *pf_tokens = nullptr;
    float *d_part_val = nullptr;  // classifier launch's per-block argmax candidates
    int *d_part_idx = nullptr;
    int n_part = 0;               // 0: argmax scans the logits instead
    float *d_attn_part = nullptr; // split attention: per (head, chunk) partials
    int attn_nch = 0;             // 0: one block per head at every position
    int attn_split_pos = 0;       // positions >= this use the split form (host picks the graph)
    // graphs, keyed by the weights they were captured with
    const l2z_weights *graph_w = nullptr;
    hipGraphExec_t g_forward[2] = {nullptr, nullptr}, g_step[2] = {nullptr, nullptr};  // [split?]
    bool use_graphs = true;
    int host_pos = 0;   // next position the greedy loop will run
    bool done = false;  // greedy loop saw BOS
    std::vector<int32_t> h_prompt;  // host copy of the greedy loop's prompt (prefill path)
    // peer-write transport: device copies of the four gathers' descriptions (xb, x, hb, logits) for
    // the kernels that push their outputs to the peers themselves (MatvecArgs::push)
    P2pArgs *d_push = nullptr;
    int max_blocks = 0;
};

namespace {

void carve_local(l2z_weights *w, const std::vector<TensorDesc> &tt)
{
    // local layout: same tensor order, each tensor (layers, rows_loc, cols); SKIP kept only
    // in file layout
    size_t off = 0;
    const float *base = w->blob;
    for (const auto &d : tt) {
        size_t r0, r1;
        shard_rows(d, w->sh, &r0, &r1);
        const bool present = w->file_layout || d.kind != SKIP;
        const float *p = base + off;
        const std::string n = d.name;
        if (n == "token_embedding_table") w->tok_emb = p;
        else if (n == "rms_att_weight") w->rms_att = p;
        else if (n == "wq") w->wq = p;
        else if (n == "wk") w->wk = p;
        else if (n == "wv") w->wv = p;
        else if (n == "wo") w->wo = p;
        else if (n == "rms_ffn_weight") w->rms_ffn = p;
        else if (n == "w1") w->w1 = p;
        else if (n == "w2") w->w2 = p;
        else if (n == "w3") w->w3 = p;
        else if (n == "rms_final_weight") w->rms_final = p;
        else if (n == "wcls") w->wcls = p;
        if (present) off += d.layers * (r1 - r0) * d.cols;
    }
    if (w->shared) w->wcls = w->tok_emb + (size_t)w->sh.v0 * w->cfg.dim;  // main.zig:112
}

size_t local_floats(const l2z_weights *w, const std::vector<TensorDesc> &tt)
{
    size_t off = 0;
    for (const auto &d : tt) {
        size_t r0, r1;
        shard_rows(d, w->sh, &r0, &r1);
        if (w->file_layout || d.kind != SKIP) off += d.layers * (r1 - r0) * d.cols;
    }
    return off;
}

int weights_alloc(const l2z_config *config, int shared_weights, const l2z_comm *comm,
                  l2z_weights **out, std::vector<TensorDesc> *tt_out)
{
    L2Z_CHECK(config != nullptr && out != nullptr, L2Z_ERR_INVALID, "weights_init: null argument");
    const int dev = current_device_for(comm);
    L2Z_TRY(ensure_device(dev));
    Shard sh;
    L2Z_TRY(make_shard(*config, comm, &sh));
    l2z_weights *w = new l2z_weights();
    w->cfg = *config;
    w->shared = shared_weights ? 1 : 0;
    w->device = dev;
    w->sh = sh;
    w->file_layout = sh.world == 1;
    *tt_out = tensor_table(*config, w->shared != 0);
    w->blob_floats = local_floats(w, *tt_out);
    hipError_t e = hipMalloc(&w->blob, w->blob_floats * sizeof(float));
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) for weights failed: %s", w->blob_floats * sizeof(float),
                  hipGetErrorString(e));
        delete w;
        return e == hipErrorOutOfMemory ? L2Z_ERR_OOM : L2Z_ERR_HIP;
    }
    carve_local(w, *tt_out);
    *out = w;
    return L2Z_OK;
}

}  // namespace

extern "C" int l2z_abi_version(void) { return L2Z_ABI_VERSION; }
extern "C" const char *l2z_last_error(void) { return l2z::g_err; }

extern "C" int l2z_device_count(int *out_n)
{
    L2Z_CHECK(out_n != nullptr, L2Z_ERR_INVALID, "l2z_device_count: null out");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) n = 0;
    *out_n = n;
    return L2Z_OK;
}

extern "C" int l2z_device_info(int dev, char *name, size_t cap, int *out_cus, uint64_t *out_hbm)
{
    L2Z_TRY(ensure_device(dev));
    hipDeviceProp_t p;
    L2Z_HIP(hipGetDeviceProperties(&p, dev));
    // some boxes report an empty marketing name
    if (name && cap) snprintf(name, cap, "%s (%s)", p.name[0] ? p.name : "AMD GPU", p.gcnArchName);
    if (out_cus) *out_cus = p.multiProcessorCount;
    if (out_hbm) *out_hbm = (uint64_t)p.totalGlobalMem;
    return L2Z_OK;
}

// src/main.zig:73 Weights.init
extern "C" int l2z_weights_init(const l2z_config *config, const float *data, size_t n_floats,
                                int shared_weights, const l2z_comm *comm, l2z_weights **out)
{
    L2Z_CHECK(data != nullptr, L2Z_ERR_INVALID, "l2z_weights_init: null data");
    std::vector<TensorDesc> tt;
    l2z_weights *w = nullptr;
    L2Z_TRY(weights_alloc(config, shared_weights, comm, &w, &tt));
    const size_t need = tt.back().offset + tt.back().count();
    if (n_floats < need) {
        set_error("l2z_weights_init: blob has %zu f32, config needs %zu", n_floats, need);
        l2z_weights_free(w);
        return L2Z_ERR_INVALID;
    }
    hipError_t e = hipSuccess;
    if (w->file_layout) {
        // one allocation, byte-identical to the file blob; copy in 256 MiB pieces
        const size_t piece = (size_t)64 << 20;
        for (size_t o = 0; o < need && e == hipSuccess; o += piece) {
            const size_t n = need - o < piece ? need - o : piece;
            e = hipMemcpy(w->blob + o, data + o, n * sizeof(float), hipMemcpyHostToDevice);
        }
    } else {
        // sharded direct upload: only this rank's rows are read from the host blob
        size_t off = 0;
        for (const auto &d : tt) {
            if (d.kind == SKIP) continue;
            size_t r0, r1;
            shard_rows(d, w->sh, &r0, &r1);
            const size_t rl = r1 - r0;
            for (size_t l = 0; l < d.layers && e == hipSuccess; l++) {
                const float *src = data + d.offset + (l * d.rows + r0) * d.cols;
                e = hipMemcpy(w->blob + off + l * rl * d.cols, src, rl * d.cols * sizeof(float),
                              hipMemcpyHostToDevice);
            }
            off += d.layers * rl * d.cols;
        }
    }
    if (e != hipSuccess) {
        set_error("weight upload failed: %s", hipGetErrorString(e));
        l2z_weights_free(w);
        return L2Z_ERR_HIP;
    }
    *out = w;
    return L2Z_OK;
}

extern "C" int l2z_weights_init_synthetic(const l2z_config *config, int shared_weights,
                                          uint64_t seed, const l2z_comm *comm, l2z_weights **out)
{
    std::vector<TensorDesc> tt;
    l2z_weights *w = nullptr;
    L2Z_TRY(weights_alloc(config, shared_weights, comm, &w, &tt));
    hipError_t e = hipSuccess;
    size_t off = 0;
    for (const auto &d : tt) {
        if (d.kind == SKIP && !w->file_layout) continue;
        size_t r0, r1;
        shard_rows(d, w->sh, &r0, &r1);
        const size_t rl = r1 - r0;
        for (size_t l = 0; l < d.layers && e == hipSuccess; l++) {
            const uint64_t base = d.offset + (l * d.rows + r0) * d.cols;
            e = launch_synth_fill(w->blob + off + l * rl * d.cols, base, rl * d.cols, seed, d.scale,
                                  d.bias, nullptr);
        }
        off += d.layers * rl * d.cols;
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        set_error("synthetic weight fill failed: %s", hipGetErrorString(e));
        l2z_weights_free(w);
        return L2Z_ERR_HIP;
    }
    *out = w;
    return L2Z_OK;
}

extern "C" int l2z_weights_read(const l2z_weights *w, size_t offset, size_t count, float *out)
{
    L2Z_CHECK(w != nullptr && out != nullptr, L2Z_ERR_INVALID, "l2z_weights_read: null argument");
    L2Z_CHECK(w->file_layout, L2Z_ERR_INVALID, "l2z_weights_read: only for unsharded weights");
    L2Z_CHECK(offset + count <= w->blob_floats, L2Z_ERR_INVALID, "l2z_weights_read: out of range");
    L2Z_HIP(hipSetDevice(w->device));
    L2Z_HIP(hipMemcpy(out, w->blob + offset, count * sizeof(float), hipMemcpyDeviceToHost));
    return L2Z_OK;
}

extern "C" void l2z_weights_free(l2z_weights *w)
{
    if (!w) return;
    if (w->blob) (void)hipFree(w->blob);
    delete w;
}

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
    s->max_blocks = 8;  // per CU; the launcher also caps at the occupancy query
    if (const char *e = getenv("L2Z_MAX_BLOCKS_PER_CU")) {
        const int v = atoi(e);
        if (v > 0) s->max_blocks = v;
    }
    if (const char *e = getenv("L2Z_NO_GRAPH")) s->use_graphs = atoi(e) == 0;
    // Peer-write gathers are plain kernels: captured with the rest of the step.  With RCCL only,
    // the launches and collectives go out eagerly by default (multi-rank capture of RCCL calls
    // could not be exercised on the 1-GPU dev box); L2Z_COMM_GRAPH=1 captures them too (works
    // with a 1-rank communicator).  Emulated ranks are driven stage by stage, never captured.
    if (comm && comm->world > 1 && !comm->nccl && !comm->p2p) s->use_graphs = false;
    if (comm && comm->nccl && !comm->p2p) {
        const char *e = getenv("L2Z_COMM_GRAPH");
        s->use_graphs = e && atoi(e) == 1;
    }

    const size_t kv = (size_t)c.n_layers * c.seq_len * sh.kvd_loc;
    hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    auto alloc = [&](void **p, size_t bytes) {
        if (e == hipSuccess) e = hipMalloc(p, bytes ? bytes : 4);
        if (e == hipSuccess) e = hipMemset(*p, 0, bytes ? bytes : 4);
    };
    alloc((void **)&s->x, (size_t)c.dim * 4);
    alloc((void **)&s->xb, (size_t)c.dim * 4);
    alloc((void **)&s->hb, (size_t)c.hidden_dim * 4);
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
        // wins and keeps winning (2.4x at pos 2047 on the 7B shape).  The host knows pos, so it
        // replays one of two captured graphs.  L2Z_ATTN_SPLIT: 0 = never, n = n chunks at every
        // position (tests); L2Z_ATTN_SPLIT_POS moves the switch-over.
        const char *ev = getenv("L2Z_ATTN_SPLIT");
        const int mode = ev ? atoi(ev) : -1;
        // The chunk count is part of the arithmetic (the combine rounds per chunk), so it is taken
        // from the model's TOTAL head count, not this rank's share: sharded and unsharded runs then
        // use the same chunks and stay bit-identical beyond pos 256 as well.
        int nch = mode > 0 ? mode : attention_split_chunks(c.n_heads, g_cus);
        if (nch > 16) nch = 16;
        s->attn_split_pos = mode > 0 ? 0 : 256;
        if (const char *ep = getenv("L2Z_ATTN_SPLIT_POS")) s->attn_split_pos = atoi(ep);
        if (mode == 0 || c.seq_len <= s->attn_split_pos) nch = 0;
        if (nch > 1) {
            s->attn_nch = nch;
            alloc((void **)&s->d_attn_part, attention_split_part_floats(sh.heads_loc, sh.hs, nch) * 4);
        }
    }
    if (comm && comm->p2p && comm->world > 1) {
        P2pArgs t[4];
        comm_p2p_args(comm, s->xb, (size_t)sh.dim_loc, &t[0]);
        comm_p2p_args(comm, s->x, (size_t)sh.dim_loc, &t[1]);
        comm_p2p_args(comm, s->hb, (size_t)sh.hid_loc, &t[2]);
        comm_p2p_args(comm, s->logits, (size_t)sh.v_loc, &t[3]);
        alloc((void **)&s->d_push, sizeof t);
        if (e == hipSuccess) e = hipMemcpy(s->d_push, t, sizeof t, hipMemcpyHostToDevice);
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
    for (int v = 0; v < 2; v++) {
        if (s->g_forward[v]) (void)hipGraphExecDestroy(s->g_forward[v]);
        if (s->g_step[v]) (void)hipGraphExecDestroy(s->g_step[v]);
    }
    void *ptrs[] = {s->x, s->xb, s->hb, s->q, s->logits, s->key_cache, s->value_cache, s->rope,
                    s->d_token, s->d_pos, s->d_prompt, s->d_n_prompt, s->d_out_tokens, s->d_argmax,
                    s->d_part_val, s->d_part_idx, s->d_attn_part, s->pf_x, s->pf_xn, s->pf_q,
                    s->pf_att, s->pf_h1, s->pf_tokens, s->d_push};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

// ---------------------------------------------------------------------------
namespace {

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
    return L2Z_OK;
}

// The forward pass (main.zig:285-430) as 5 launches per layer + classifier
// (+ argmax/hand-over).  Token and pos are read from device memory.
// `only_stage` >= 0 runs just the launches between two gather points (and no collective):
// the single-process multi-rank emulation (l2z_emu_transformer) interleaves the ranks
// stage by stage and performs the gathers itself.  Stages: 4 per layer (after attention,
// wo, ffn13, ffn2), then the classifier, then argmax.
int enqueue_forward(l2z_runstate *s, const l2z_weights *w, bool with_step, Prof *prof,
                    int only_stage, bool split)
{
    const l2z_config &c = s->cfg;
    const Shard &sh = s->sh;
    hipStream_t st = s->stream;
    const size_t dim = c.dim, hid = c.hidden_dim;
    const int mb = s->max_blocks;
    int stage = 0;
    auto want = [&]() { return only_stage < 0 || only_stage == stage; };
    // Peer-write transport: kernels that can, store their outputs as LL words straight into the
    // peers' slots (the values travel while the launch still runs); the gather that follows then
    // only collects.  `pushed` = the launch just made did.  Not while profiling (the gather's
    // share would be hidden in the kernel's time), not for emulated ranks, not with L2Z_COMM=rccl.
    static const bool push_env = !(getenv("L2Z_P2P_PUSH") && atoi(getenv("L2Z_P2P_PUSH")) == 0) &&
                                 !(getenv("L2Z_COMM") && strcmp(getenv("L2Z_COMM"), "rccl") == 0);
    const bool can_push = push_env && s->d_push != nullptr && prof == nullptr && only_stage < 0;
    bool pushed = false;
    auto gather = [&](float *buf, size_t count_per_rank) -> int {
        stage++;
        if (only_stage >= 0) return L2Z_OK;
        if (pushed) {
            pushed = false;
            return comm_allgather_inplace_pushed(s->comm, buf, count_per_rank, st);
        }
        return comm_allgather_inplace(s->comm, buf, count_per_rank, st);
    };
    for (int l = 0; l < c.n_layers; l++) {
        float *kc = s->key_cache + (size_t)l * c.seq_len * sh.kvd_loc;  // :354 loff
        float *vc = s->value_cache + (size_t)l * c.seq_len * sh.kvd_loc;
        if (want()) {   // rmsnorm (:305) + q,k,v (:308-320) + RoPE (:336-351) + KV write (:354-358)
            MatvecArgs a = {};
            a.w0 = w->wq + (size_t)l * sh.dim_loc * dim;
            a.w1 = w->wk + (size_t)l * sh.kvd_loc * dim;
            a.w2 = w->wv + (size_t)l * sh.kvd_loc * dim;
            a.out0 = s->q; a.out1 = kc; a.out2 = vc;
            a.rows0 = sh.dim_loc; a.rows1 = sh.kvd_loc; a.rows2 = sh.kvd_loc;
            a.pos_stride1 = sh.kvd_loc; a.pos_stride2 = sh.kvd_loc;
            a.n = c.dim; a.x = s->x; a.rms_w = w->rms_att + (size_t)l * dim;
            a.pos_ptr = s->d_pos; a.rope = s->rope; a.head_size = sh.hs; a.rope_segs = 2;
            L2Z_LAUNCH(KIND_QKV, launch_matvec(a, PRO_RMS, EPI_ROPE, mb, g_cus, st));
        }
        if (want()) {   // attention (:361-389) over the local heads
            AttnArgs a = {};
            a.q = s->q; a.kcache = kc; a.vcache = vc; a.xb = s->xb + sh.dim0;
            a.pos_ptr = s->d_pos; a.head_size = sh.hs; a.kv_dim = sh.kvd_loc;
            a.kv_mul = c.n_heads / c.n_kv_heads; a.seq_len = c.seq_len;
            if (can_push && attention_push_supported(a)) {
                a.push = s->d_push + 0;
                pushed = true;
            }
            if (split && s->attn_nch > 1 && attention_split_supported(a))
                L2Z_LAUNCH(KIND_ATTN, launch_attention_split(a, sh.heads_loc, s->attn_nch,
                                                             s->d_attn_part, st));
            else
                L2Z_LAUNCH(KIND_ATTN, launch_attention(a, sh.heads_loc, st));
        }
        L2Z_TRY(gather(s->xb, sh.dim_loc));
        if (want()) {   // wo (:392) + residual (:395)
            MatvecArgs a = {};
            a.w0 = w->wo + (size_t)l * sh.dim_loc * dim;
            a.out0 = s->x + sh.dim0; a.resid = s->x + sh.dim0;
            a.rows0 = sh.dim_loc; a.n = c.dim; a.x = s->xb;
            if (can_push) a.push = s->d_push + 1;
            L2Z_LAUNCH(KIND_WO, launch_matvec(a, PRO_NONE, EPI_RESID, mb, g_cus, st, nullptr, &pushed));
        }
        L2Z_TRY(gather(s->x, sh.dim_loc));
        if (want()) {   // rmsnorm (:398) + w1,w3 (:405-408) + SiLU*mul (:411-416)
            MatvecArgs a = {};
            a.w0 = w->w1 + (size_t)l * sh.hid_loc * dim;
            a.w1 = w->w3 + (size_t)l * sh.hid_loc * dim;
            a.out0 = s->hb + sh.hid0;
            a.rows0 = sh.hid_loc; a.rows1 = sh.hid_loc; a.n = c.dim;
            a.x = s->x; a.rms_w = w->rms_ffn + (size_t)l * dim;
            if (can_push) a.push = s->d_push + 2;
            L2Z_LAUNCH(KIND_FFN13, launch_matvec(a, PRO_RMS, EPI_SWIGLU, mb, g_cus, st, nullptr, &pushed));
        }
        L2Z_TRY(gather(s->hb, sh.hid_loc));
        if (want()) {   // w2 (:419) + residual (:422)
            MatvecArgs a = {};
            a.w0 = w->w2 + (size_t)l * sh.dim_loc * hid;
            a.out0 = s->x + sh.dim0; a.resid = s->x + sh.dim0;
            a.rows0 = sh.dim_loc; a.n = c.hidden_dim; a.x = s->hb;
            if (can_push) a.push = s->d_push + 1;
            L2Z_LAUNCH(KIND_FFN2, launch_matvec(a, PRO_NONE, EPI_RESID, mb, g_cus, st, nullptr, &pushed));
        }
        L2Z_TRY(gather(s->x, sh.dim_loc));
    }
    if (want()) {   // final rmsnorm (:426) + classifier (:429)
        MatvecArgs a = {};
        a.w0 = w->wcls; a.out0 = s->logits + sh.v0;
        a.rows0 = sh.v_loc; a.n = c.dim; a.x = s->x; a.rms_w = w->rms_final;
        a.part_val = s->d_part_val; a.part_idx = s->d_part_idx; a.row_offset = sh.v0;
        // single GPU, vector path: the launch also leaves one argmax candidate per block
        const bool fuse = sh.world == 1 && matvec_vector_width(c.dim);
        int grid = 0;
        if (can_push) a.push = s->d_push + 3;
        L2Z_LAUNCH(KIND_CLS, launch_matvec(a, PRO_RMS, fuse ? EPI_ARGMAX : EPI_STORE, mb, g_cus, st,
                                           &grid, &pushed));
        s->n_part = fuse ? grid : 0;
    }
    L2Z_TRY(gather(s->logits, sh.v_loc));
    if (with_step && want()) {
        ArgmaxArgs a = {};
        a.logits = s->logits; a.vocab = c.vocab_size; a.token_ptr = s->d_token;
        if (s->n_part > 0) { a.part_val = s->d_part_val; a.part_idx = s->d_part_idx; a.n_part = s->n_part; }
        a.pos_ptr = s->d_pos; a.prompt = s->d_prompt; a.n_prompt_ptr = s->d_n_prompt;
        a.out_tokens = s->d_out_tokens; a.argmax_out = s->d_argmax; a.tok_emb = w->tok_emb;
        a.x = s->x; a.dim = c.dim; a.advance = 1;
        L2Z_LAUNCH(KIND_ARGMAX, launch_argmax(a, st));
    }
    return L2Z_OK;
}

bool use_split(const l2z_runstate *s, int pos) { return s->attn_nch > 1 && pos >= s->attn_split_pos; }

int build_graph(l2z_runstate *s, const l2z_weights *w, bool with_step, bool split,
                hipGraphExec_t *out)
{
    hipGraph_t graph = nullptr;
    L2Z_HIP(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
    int rc = enqueue_forward(s, w, with_step, nullptr, -1, split);
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
    for (int v = 0; v < 2; v++) {
        if (s->g_forward[v]) { (void)hipGraphExecDestroy(s->g_forward[v]); s->g_forward[v] = nullptr; }
        if (s->g_step[v]) { (void)hipGraphExecDestroy(s->g_step[v]); s->g_step[v] = nullptr; }
    }
    s->graph_w = nullptr;
}

int ensure_graphs(l2z_runstate *s, const l2z_weights *w)
{
    if (!s->use_graphs) return L2Z_OK;
    if (s->graph_w == w && s->g_forward[0] && s->g_step[0]) return L2Z_OK;
    drop_graphs(s);
    const int n_var = s->attn_nch > 1 ? 2 : 1;
    int rc = L2Z_OK;
    for (int v = 0; v < n_var && rc == L2Z_OK; v++) {
        rc = build_graph(s, w, false, v == 1, &s->g_forward[v]);
        if (rc == L2Z_OK) rc = build_graph(s, w, true, v == 1, &s->g_step[v]);
    }
    if (rc != L2Z_OK) {
        // capture is an optimisation, not a requirement: run the same launches eagerly
        fprintf(stderr, "llama2_hip: hipGraph capture failed (%s); launching eagerly\n", g_err);
        drop_graphs(s);
        s->use_graphs = false;
        (void)hipGetLastError();
        return L2Z_OK;
    }
    s->graph_w = w;
    return L2Z_OK;
}

// one forward pass at position `pos` (the host mirrors the device-side pos)
int run_forward(l2z_runstate *s, const l2z_weights *w, bool with_step, int pos)
{
    const bool split = use_split(s, pos);
    L2Z_CHECK(s->sh.world == 1 || (s->comm && (s->comm->nccl || s->comm->p2p)), L2Z_ERR_STATE,
              "sharded runstate without a transport: connect the group (RCCL id or "
              "l2z_comm_p2p_export/_connect), or drive emulated ranks with l2z_emu_transformer");
    L2Z_TRY(comm_check(s->comm));
    L2Z_TRY(ensure_graphs(s, w));
    if (s->use_graphs) {
        const int v = split ? 1 : 0;
        L2Z_HIP(hipGraphLaunch(with_step ? s->g_step[v] : s->g_forward[v], s->stream));
        return L2Z_OK;
    }
    return enqueue_forward(s, w, with_step, nullptr, -1, split);
}

}  // namespace

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
    L2Z_HIP(hipMemcpyAsync(out_logits, s->logits, (size_t)s->cfg.vocab_size * sizeof(float),
                           hipMemcpyDeviceToHost, s->stream));
    L2Z_HIP(hipStreamSynchronize(s->stream));
    L2Z_TRY(comm_check(s->comm));
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
    L2Z_HIP(hipStreamSynchronize(s->stream));
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

extern "C" int l2z_synchronize(l2z_runstate *s)
{
    L2Z_CHECK(s != nullptr, L2Z_ERR_INVALID, "l2z_synchronize: null runstate");
    L2Z_HIP(hipSetDevice(s->device));
    L2Z_HIP(hipStreamSynchronize(s->stream));
    L2Z_TRY(comm_check(s->comm));
    return L2Z_OK;
}

// ---------------------------------------------------------------------------
// Batched prefill: the same state change as calling
// l2z_transformer(tokens[i], pos0 + i) for i = 0..n-1 -- KV-cache rows pos0..pos0+n-1 written in
// every layer, logits of the LAST position left in the runstate -- but every weight matrix is
// streamed once per chunk of up to kPrefillChunk tokens and multiplied on the fp32 matrix cores.
namespace {
static int prefill_chunk_tokens()
{
    static int n = 0;
    if (n == 0) {
        const char *e = getenv("L2Z_PF_CHUNK");
        n = e ? atoi(e) : 512;
        if (n < 16) n = 16;
        if (n > 2048) n = 2048;
    }
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
        L2Z_HIP(launch_prefill_gemm(PG_STORE, s->pf_xn, dim, w->w1 + (size_t)l * hid * dim, s->pf_h1, hid,
                                    P, hid, dim, pos0, s->rope, hs, st));                   // :405
        L2Z_HIP(launch_prefill_gemm(PG_SWIGLU, s->pf_xn, dim, w->w3 + (size_t)l * hid * dim, s->pf_h1, hid,
                                    P, hid, dim, pos0, s->rope, hs, st));   // :408 + :411-416 in the epilogue
        L2Z_HIP(launch_prefill_gemm(PG_RESID, s->pf_h1, hid, w->w2 + (size_t)l * dim * hid, s->pf_x, dim,
                                    P, dim, hid, pos0, s->rope, hs, st));                   // :419-422
    }
    return L2Z_OK;
}
}  // namespace

constexpr int kPrefillMinPrompt = L2Z_PREFILL_MIN_PROMPT;  // shorter prompts: the stepped loop is as fast

static bool prefill_enabled()
{
    static int on = -1;
    if (on < 0) {
        const char *e = getenv("L2Z_PREFILL");
        on = (e && atoi(e) == 0) ? 0 : 1;
    }
    return on == 1;
}

static int prefill_check(const l2z_config *config, const l2z_runstate *s)
{
    L2Z_CHECK(s->sh.world == 1, L2Z_ERR_INVALID, "l2z_prefill: not available on a sharded runstate");
    L2Z_CHECK(config->dim % 4 == 0 && config->hidden_dim % 4 == 0 && s->sh.hs % 4 == 0 &&
                  s->sh.hs <= 256, L2Z_ERR_INVALID,
              "l2z_prefill: needs dim, hidden_dim, head_size multiples of 4 and head_size <= 256");
    return L2Z_OK;
}

// positions pos0 .. pos0+n-1 in chunks; leaves the last position's residual row in RunState.x
static int prefill_tokens(l2z_runstate *s, const l2z_weights *w, const int32_t *tokens, int n_tokens,
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
        s->sh.world == 1 && config->dim % 4 == 0 && config->hidden_dim % 4 == 0 &&
        s->sh.hs % 4 == 0 && s->sh.hs <= 256 &&
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
    int rc = enqueue_forward(s, w, true, &prof, -1, use_split(s, pos));
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
    const int n_stages = 4 * c.n_layers + 1;
    for (int stage = 0; stage < n_stages; stage++) {
        for (int r = 0; r < n_ranks; r++)
            L2Z_TRY(enqueue_forward(ss[r], ws[r], false, nullptr, stage, use_split(ss[r], pos)));
        for (int r = 0; r < n_ranks; r++) L2Z_HIP(hipStreamSynchronize(ss[r]->stream));
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

// ---------------------------------------------------------------------------
// Kernel-level test hooks: upload, run the SAME device code the forward pass
// uses, download.
namespace {
struct DevBuf {
    float *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { L2Z_HIP(hipMalloc(&p, (n ? n : 1) * sizeof(float))); return L2Z_OK; }
    int up(const float *h, size_t n) { L2Z_HIP(hipMemcpy(p, h, n * sizeof(float), hipMemcpyHostToDevice)); return L2Z_OK; }
    int down(float *h, size_t n) { L2Z_HIP(hipMemcpy(h, p, n * sizeof(float), hipMemcpyDeviceToHost)); return L2Z_OK; }
};
}  // namespace

extern "C" int l2z_matmul_fused(int N, float *const *outs, const float *x, const float *const *ws,
                                size_t n, size_t d)
{
    L2Z_CHECK(N >= 1 && N <= kMaxSeg && outs && x && ws && n > 0 && d > 0, L2Z_ERR_INVALID,
              "l2z_matmul_fused: bad arguments");
    L2Z_CHECK(n < (1u << 30) && d < (1u << 30), L2Z_ERR_INVALID, "l2z_matmul_fused: too large");
    L2Z_TRY(ensure_device(current_device_for(nullptr)));
    DevBuf dx, dw[kMaxSeg], dout[kMaxSeg];
    L2Z_TRY(dx.alloc(n));
    L2Z_TRY(dx.up(x, n));
    MatvecArgs a = {};
    a.n = (int)n; a.x = dx.p;
    for (int j = 0; j < N; j++) {
        L2Z_TRY(dw[j].alloc(n * d));
        L2Z_TRY(dw[j].up(ws[j], n * d));
        L2Z_TRY(dout[j].alloc(d));
    }
    a.w0 = dw[0].p; a.out0 = dout[0].p; a.rows0 = (int)d;
    if (N > 1) { a.w1 = dw[1].p; a.out1 = dout[1].p; a.rows1 = (int)d; }
    if (N > 2) { a.w2 = dw[2].p; a.out2 = dout[2].p; a.rows2 = (int)d; }
    L2Z_HIP(launch_matvec(a, PRO_NONE, EPI_STORE, 8, g_cus, nullptr));
    L2Z_HIP(hipDeviceSynchronize());
    for (int j = 0; j < N; j++) L2Z_TRY(dout[j].down(outs[j], d));
    return L2Z_OK;
}

extern "C" int l2z_matmul(float *xout, const float *x, const float *w, size_t n, size_t d)
{
    float *outs[1] = {xout};
    const float *ws[1] = {w};
    return l2z_matmul_fused(1, outs, x, ws, n, d);
}

extern "C" int l2z_rmsnorm(float *o, const float *x, const float *w, size_t n)
{
    L2Z_CHECK(o && x && w && n > 0 && n < (1u << 30), L2Z_ERR_INVALID, "l2z_rmsnorm: bad arguments");
    L2Z_CHECK(matvec_lds_bytes((int)n) <= 160 * 1024, L2Z_ERR_INVALID, "l2z_rmsnorm: n too large for LDS");
    L2Z_TRY(ensure_device(current_device_for(nullptr)));
    DevBuf dx, dw, dout;
    L2Z_TRY(dx.alloc(n)); L2Z_TRY(dw.alloc(n)); L2Z_TRY(dout.alloc(n));
    L2Z_TRY(dx.up(x, n)); L2Z_TRY(dw.up(w, n));
    L2Z_HIP(launch_rmsnorm(dout.p, dx.p, dw.p, (int)n, nullptr));
    L2Z_HIP(hipDeviceSynchronize());
    return dout.down(o, n);
}

extern "C" int l2z_softmax(float *x, size_t n)
{
    L2Z_CHECK(x && n > 0 && n < (1u << 30), L2Z_ERR_INVALID, "l2z_softmax: bad arguments");
    L2Z_TRY(ensure_device(current_device_for(nullptr)));
    DevBuf dx;
    L2Z_TRY(dx.alloc(n)); L2Z_TRY(dx.up(x, n));
    L2Z_HIP(launch_softmax(dx.p, (int)n, nullptr));
    L2Z_HIP(hipDeviceSynchronize());
    return dx.down(x, n);
}

extern "C" int l2z_vector_dot_product(float *out, const float *x, const float *y, size_t n)
{
    L2Z_CHECK(out && x && y && n > 0 && n < (1u << 30), L2Z_ERR_INVALID, "l2z_vector_dot_product: bad arguments");
    L2Z_TRY(ensure_device(current_device_for(nullptr)));
    DevBuf dx, dy, dout;
    L2Z_TRY(dx.alloc(n)); L2Z_TRY(dy.alloc(n)); L2Z_TRY(dout.alloc(1));
    L2Z_TRY(dx.up(x, n)); L2Z_TRY(dy.up(y, n));
    L2Z_HIP(launch_dot(dout.p, dx.p, dy.p, (int)n, nullptr));
    L2Z_HIP(hipDeviceSynchronize());
    return dout.down(out, 1);
}

extern "C" int l2z_vector_weighted_sum_rows(float *xout, size_t xout_len, const float *rows,
                                            size_t rows_len, size_t row_stride,
                                            const float *weights, size_t n_weights)
{
    L2Z_CHECK(xout && rows && weights && xout_len > 0 && n_weights > 0, L2Z_ERR_INVALID,
              "l2z_vector_weighted_sum_rows: bad arguments");
    // main.zig:660-661 asserts
    L2Z_CHECK(row_stride >= xout_len && rows_len >= (n_weights - 1) * row_stride + xout_len,
              L2Z_ERR_INVALID, "l2z_vector_weighted_sum_rows: stride/length contract violated");
    L2Z_TRY(ensure_device(current_device_for(nullptr)));
    DevBuf dr, dw, dout;
    L2Z_TRY(dr.alloc(rows_len)); L2Z_TRY(dw.alloc(n_weights)); L2Z_TRY(dout.alloc(xout_len));
    L2Z_TRY(dr.up(rows, rows_len)); L2Z_TRY(dw.up(weights, n_weights));
    L2Z_HIP(launch_weighted_sum_rows(dout.p, (int)xout_len, dr.p, (int)row_stride, dw.p,
                                     (int)n_weights, nullptr));
    L2Z_HIP(hipDeviceSynchronize());
    return dout.down(xout, xout_len);
}

extern "C" int l2z_argmax_host(const float *x, size_t n, size_t *out_index)
{
    L2Z_CHECK(x && out_index && n > 0 && n < (1u << 30), L2Z_ERR_INVALID, "l2z_argmax_host: bad arguments");
    L2Z_TRY(ensure_device(current_device_for(nullptr)));
    DevBuf dx;
    int *didx = nullptr;
    L2Z_TRY(dx.alloc(n)); L2Z_TRY(dx.up(x, n));
    L2Z_HIP(hipMalloc(&didx, sizeof(int)));
    ArgmaxArgs a = {};
    a.logits = dx.p; a.vocab = (int)n; a.argmax_out = didx; a.advance = 0;
    hipError_t e = launch_argmax(a, nullptr);
    int idx = 0;
    if (e == hipSuccess) e = hipMemcpy(&idx, didx, sizeof(int), hipMemcpyDeviceToHost);
    (void)hipFree(didx);
    L2Z_HIP(e);
    *out_index = (size_t)idx;
    return L2Z_OK;
}

#ifdef L2Z_DBG_TS
namespace l2z { hipError_t dbg_ts_read(long long *out); }
extern "C" int l2z_dbg_ts(long long *out) { return l2z::dbg_ts_read(out) == hipSuccess ? 0 : -3; }
#endif
