// l2z_state.h -- the device-resident Weights / RunState objects behind the C ABI and the internal
// functions the translation units share (weights.cpp, runstate.cpp, forward.cpp, prefill_host.cpp,
// hooks.cpp).  Product code; nothing under oracle/ is referenced.
#pragma once
#include <cstdint>
#include <vector>

#include "l2z_comm.h"
#include "l2z_internal.h"
#include "tunables.h"

namespace l2z {

// ----- shard geometry (DESIGN.md "Sharding"; world == 1 -> everything local) -----
struct Shard {
    int rank = 0, world = 1;
    int hs = 0;        // head_size
    int dim0 = 0, dim_loc = 0;   // rows of q / wo / w2 and slice of x, xb owned here
    int kvd_loc = 0;             // local kv_dim (whole kv heads)
    int heads_loc = 0;
    int hid0 = 0, hid_loc = 0;   // rows of w1/w3, slice of hb
    int v0 = 0, v_loc = 0;       // rows of wcls, slice of logits
    // Scheme B (L2Z_SCHEME_B, world > 1): Wo and W2 are sharded by COLUMNS -- this rank holds the
    // columns its own heads / hidden rows feed, [dim][dimc_pad] and [dim][hidc_pad] per layer, the pad columns zero --
    // and produces a partial [dim] vector that an all-reduce sums.  dimc_pad / hidc_pad: dim_loc / hid_loc rounded
    // up to a width the vector mat-vec takes (pad_cols)
    bool scheme_b = false;
    int dimc_pad = 0, hidc_pad = 0;
};
int pad_cols(int n);  // widths above 768 floats: multiples of 256 (a row = whole 64-lane sweeps of 16 bytes); below: of 4
int make_shard(const l2z_config &c, const l2z_comm *comm, Shard *out);

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
std::vector<TensorDesc> tensor_table(const l2z_config &c, bool shared);
void shard_rows(const TensorDesc &d, const Shard &s, size_t *r0, size_t *r1);

extern int g_cus;  // compute units of the device in use (set by ensure_device)
int ensure_device(int device);
int current_device_for(const l2z_comm *comm);

}  // namespace l2z

struct l2z_weights {
    uint64_t uid = 0;  // unique per object for the life of the process: keys a runstate's captured graphs
    l2z_config cfg;
    int shared;
    int device;
    l2z::Shard sh;
    // one allocation: the tensors of the file in file order (this rank's rows of each), EXCEPT that W1 and W3
    // share one slot where W1 stood, row-interleaved -- [layer][row][W1 row | W3 row] -- so that the pair of rows
    // a feed-forward output needs is one contiguous run; W3's own slot is empty
    float *blob = nullptr;
    size_t blob_floats = 0;
    bool file_layout = false;    // world == 1: every tensor of the file is resident (the unread freq_cis tables too)
    std::vector<size_t> dev_off; // device offset (floats) of each tensor of the table; W3: that of the shared slot + cols
    // carved device pointers (local shards when world > 1)
    const float *tok_emb = nullptr, *rms_att = nullptr, *rms_ffn = nullptr, *rms_final = nullptr;
    const float *wq = nullptr, *wk = nullptr, *wv = nullptr, *wo = nullptr;
    // w1 / w3: row 0 of W1 / of W3 in the shared slot; consecutive rows of either are 2 * dim floats apart
    const float *w1 = nullptr, *w2 = nullptr, *w3 = nullptr, *wcls = nullptr;
};

enum { KIND_QKV = 0, KIND_ATTN, KIND_WO, KIND_FFN13, KIND_FFN2, KIND_CLS, KIND_ARGMAX, KIND_GATHER, KIND_COUNT };
static const char *const kKindNames[KIND_COUNT] = {"qkv", "attn", "wo", "ffn13", "ffn2", "cls", "argmax", "gather"};
static_assert(KIND_COUNT == L2Z_N_KINDS, "include/llama2_hip_test.h L2Z_N_KINDS");

struct l2z_runstate {
    l2z_config cfg;
    int device;
    l2z::Shard sh;
    const l2z_comm *comm = nullptr;
    hipStream_t stream = nullptr;
    // main.zig:119-135 (k, v, xb2, hb2, logits_indexed have no device twin:
    // k/v go straight into the cache rows, xb2/hb2 are fused away)
    float *x = nullptr, *xb = nullptr, *hb = nullptr, *q = nullptr, *logits = nullptr;
    float *part = nullptr;    // scheme B: this rank's partial [dim] output of wo / w2 (rank 0's includes the residual), summed by the all-reduce
    float *key_cache = nullptr, *value_cache = nullptr;
    float2 *rope = nullptr;  // (seq_len, head_size/2) {cos, sin}
    // loop state on the device
    int *d_token = nullptr, *d_pos = nullptr, *d_prompt = nullptr, *d_n_prompt = nullptr;
    int *d_out_tokens = nullptr, *d_argmax = nullptr;
    // batched prefill scratch (allocated on first l2z_prefill): [kPrefillChunk, dim|hidden]
    float *pf_x = nullptr, *pf_xn = nullptr, *pf_q = nullptr, *pf_att = nullptr;
    float *pf_h1 = nullptr;
    int pf_ld_xn = 0, pf_ld_att = 0, pf_ld_h1 = 0;   // row pitches of pf_xn / pf_att / pf_h1 (prefill_host.cpp pf_ld: zero-padded)
    float *pf_stage = nullptr;  // sharded: [world][P, n_loc] blocks of the matrix being gathered
    float *pf_part = nullptr;   // scheme B: this rank's partial [P, dim] product of its column shard of Wo / W2
    int *pf_tokens = nullptr;
    l2z::DeferredSum pf_pending = {nullptr, 1, 128, 1, false};   // K ranges' sums of Wo / W2 the next rmsnorm launch adds (prefill_host.cpp)
    int pf_planes_att = 0, pf_planes_h1 = 0;   // PLANES_*: whether the attention output's / the gated rows' planes stand (prefill_host.cpp)
    l2z::SplitKWs pf_sk = {nullptr, nullptr, 0, 0, nullptr, 0, nullptr, 0};  // split-K workspace of the tile GEMM (chunks of <= 256 tokens)
    int pf_cap = 0;             // tokens per chunk the scratch above was allocated for
    float *d_probs = nullptr;     // l2z_probs_read: softmax(logits / temperature), allocated on first use
    float *h_stage = nullptr;     // ... and its pinned host landing buffer
    float *d_part_val = nullptr;  // classifier launch's per-block argmax candidates
    int *d_part_idx = nullptr;
    int n_part = 0;               // 0: argmax scans the logits instead
    int n_part_step = 0, n_part_fwd = 0;  // ... as left by the captured step / forward graphs (a shard's differ)
    float *d_attn_part = nullptr; // split attention: per (head, chunk) partials
    int *d_attn_cnt = nullptr;    // split attention: arrivals per local head (zero between launches)
    int attn_nch = 0;             // 0: one block per head at every position
    int attn_split_pos = 0;       // positions >= this use the split form (host picks the graph)
    int attn_short_pos = 0;       // positions < this use the 256-thread speculative one-block-per-head form
    int attn_split_wide_pos = 0;  // split form: 256 threads per block below this position, 1024 from it on
    // graphs, keyed by the uid of the weights they were captured with (a freed object's address
    // is commonly handed to the next one)
    uint64_t graph_w_uid = 0;
    hipGraphExec_t g_forward[4] = {nullptr, nullptr, nullptr, nullptr}, g_step[4] = {nullptr, nullptr, nullptr, nullptr};  // [attention variant]
    bool use_graphs = true;
    int host_pos = 0;   // next position the greedy loop will run
    bool done = false;  // greedy loop saw BOS
    std::vector<int32_t> h_prompt;  // host copy of the greedy loop's prompt (prefill path)
    // peer-write transport: device copies of the four gathers' descriptions (xb, x, hb, logits) for
    // the kernels that push their outputs to the peers themselves (MatvecArgs::push)
    l2z::P2pArgs *d_push = nullptr;
    int n_gathers = 0;        // collectives per forward pass at world > 1: 4 gathers per layer + logits (scheme B: 2 all-reduces per layer + logits)
    bool ll_consume = false;  // peer-write transport, consumer side: mat-vecs read their gathered input
                              // as LL words from the landing slot; no gather launch except the logits
    // Greedy steps of a shard group on the peer-write transport end in a candidate exchange (ArgmaxArgs::xchg) instead
    // of the logits gather: after such a step `logits` holds only this rank's rows, and the first call that reads the
    // whole vector (l2z_logits_read, l2z_probs_read, l2z_argmax, l2z_runstate_read) gathers it -- a COLLECTIVE then:
    // every rank of the group makes it, as every rank makes every other call
    bool xchg_steps = false;
    bool logits_partial = false;
    bool fused_qkv_attn = false;  // small models: qkv + RoPE + KV write + attention in one launch
    int max_blocks = 0;
    int tl_attn_seq = 0;           // attention launches enqueued so far (AttnArgs::tl_seq, measurement builds)
};

namespace l2z {

// forward.cpp
struct Prof;
int check_pair(const l2z_config *config, const l2z_runstate *s, const l2z_weights *w);
// attention variant of a position (the host knows pos and replays the graph captured for its variant):
// 0 short context (256-thread speculative form), 1 one block per head as the shape picks, 2 split with 256
// threads per block, 3 split with 1024
enum { ATTN_SHORT = 0, ATTN_HEAD = 1, ATTN_SPLIT_S = 2, ATTN_SPLIT = 3, ATTN_VARIANTS = 4 };
int enqueue_forward(l2z_runstate *s, const l2z_weights *w, bool with_step, Prof *prof, int only_stage,
                    int variant, int only_kind = -1);
int attn_variant(const l2z_runstate *s, int pos);
int run_forward(l2z_runstate *s, const l2z_weights *w, bool with_step, int pos);
int ensure_logits(l2z_runstate *s);  // gathers the logits if the last pass left only this rank's rows (see xchg_steps)
void drop_graphs(l2z_runstate *s);

// prefill_host.cpp
constexpr int kPrefillMinPrompt = L2Z_PREFILL_MIN_PROMPT;  // shorter prompts: the stepped loop is as fast
bool prefill_enabled();
bool prefill_usable(const l2z_runstate *s);
bool prefill_shard_takes_the_unsharded_kernels(const l2z_config &c, const Shard &sh);
int prefill_check(const l2z_config *config, const l2z_runstate *s);
int prefill_tokens(l2z_runstate *s, const l2z_weights *w, const int32_t *tokens, int n_tokens, int pos0);

}  // namespace l2z
