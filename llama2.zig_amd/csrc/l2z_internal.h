// l2z_internal.h -- shared declarations for the HIP forward-pass library.
// Product code: never includes anything under oracle/.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/llama2_hip.h"
#include "../../include/llama2_hip_test.h"

namespace l2z {

struct P2pArgs;  // l2z_comm.h: device-side description of one peer-write gather

// By value in a consumer's arguments (sharded runs, peer-write transport): where it finds its input
// vector as LL words {value, epoch} -- this rank's own landing slots, filled by every rank's
// producers (p2p.hip) -- instead of a plain buffer.  slots == null: plain buffer.
struct LLIn {
    const unsigned long long *slots;  // own arena + reserved head
    unsigned slot_floats;
    unsigned count;       // floats per rank (names the late peer when a wait times out)
    int gi;               // index of the gather that fills it (epoch = ctl[0] + gi)
    int *ctl;             // l2z_comm::d_ctl
    int *h_err;           // l2z_comm::h_err
    long long timeout_ticks;
};

void set_error(const char *fmt, ...);

#define L2Z_HIP(expr)                                                                    \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            ::l2z::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),      \
                             __FILE__, __LINE__);                                        \
            return _e == hipErrorOutOfMemory ? L2Z_ERR_OOM                               \
                   : (_e == hipErrorNoDevice || _e == hipErrorInvalidDevice)             \
                       ? L2Z_ERR_NO_DEVICE                                               \
                       : L2Z_ERR_HIP;                                                    \
        }                                                                                \
    } while (0)

#define L2Z_CHECK(cond, code, ...)          \
    do {                                    \
        if (!(cond)) {                      \
            ::l2z::set_error(__VA_ARGS__);  \
            return (code);                  \
        }                                   \
    } while (0)

#define L2Z_TRY(expr)              \
    do {                           \
        int _s = (expr);           \
        if (_s != L2Z_OK) return _s; \
    } while (0)

// ---------------------------------------------------------------------------
// Kernel argument blocks (passed by value; all pointers are device pointers)
// ---------------------------------------------------------------------------

enum Prologue { PRO_NONE = 0, PRO_RMS = 1 };
enum Epilogue { EPI_STORE = 0, EPI_ROPE = 1, EPI_RESID = 2, EPI_SWIGLU = 3, EPI_ARGMAX = 4 };

constexpr int kMaxSeg = 3;

// main.zig:715 argmax + main.zig:999-1000,1036 hand-over to the next step
struct ArgmaxArgs {
    const float *logits;
    int vocab;
    const float *part_val;   // if non-null: scan n_part (value, index) candidates instead
    const int *part_idx;
    int n_part;
    int *token_ptr;          // in/out: current token
    int *pos_ptr;            // in/out
    const int *prompt;       // forced tokens (n_prompt)
    const int *n_prompt_ptr;
    int *out_tokens;         // out_tokens[pos] = next
    int *argmax_out;         // plain argmax result
    const float *tok_emb;    // (vocab, dim): next step's embedding row -> x
    float *x;
    int dim;
    int advance;             // 1: greedy step (write token/pos/x), 0: argmax only
    int *epoch_ctl;          // this launch closes the pass of a shard group: ctl[kCtlEpoch] += epoch_add (null: not)
    int epoch_add;
    // Shard group on the peer-write transport, greedy step ("a local-argmax + N-pair exchange"): the
    // candidates are this rank's vocabulary rows only; the ranks exchange their (max, first index) pairs as LL words in the
    // landing slots (hand-over xchg_gi of the pass, words 2r and 2r + 1 of rank r) and every rank picks the same winner --
    // larger value, equal values: lower index (main.zig:720) -- instead of gathering 32000 logits to scan them.  null: not
    const P2pArgs *xchg;
    int xchg_gi;
};

// One fused mat-vec launch: up to 3 row-major (rowsJ, n) matrices sharing x.
// main.zig:530 matmul_fused(N) -- plus what the reference does right before
// (rmsnorm, :305/:398/:426) and right after (RoPE+KV write :336-358, accum
// :395/:422, SiLU*mul :411-416) each call.  Named scalar fields on purpose:
// arrays here get indexed dynamically by the optimiser and land in scratch.
struct MatvecArgs {
    // unused segments: null, rows = 0.  EPI_SWIGLU: w0 is W1 | W3 ROW-INTERLEAVED (row 2p = W1 row p, row 2p + 1 =
    // W3 row p: the device layout of the weights), rows0 = rows1 = pairs, w1 is not read
    const float *w0, *w1, *w2;
    float *out0, *out1, *out2;
    int rows0, rows1, rows2;
    int pos_stride1, pos_stride2;  // out1/out2 += pos * stride (row select of a flat (rows, stride) buffer)
    // EPI_ROPE into the device KV cache: out1 / out2 are one layer's head-major caches
    // [kv_heads][seq_len][head_size]; row r of the segment goes to (r / head_size) * kv_head_stride +
    // pos * head_size + r % head_size.  0: flat form above.
    size_t kv_head_stride;
    int n;                    // columns = length of x
    const float *x;
    const float *rms_w;       // PRO_RMS: rmsnorm weight (n)
    const float *resid;       // EPI_RESID: out = resid + W.x (may alias out0)
    const int *pos_ptr;       // EPI_ROPE: device int, current position
    const float2 *rope;       // (seq_len, head_size/2) {cos, sin}
    int head_size;
    int rope_segs;            // leading segments that get rotated (q, k -> 2)
    // EPI_ARGMAX (classifier): store logits AND one (max, first index) per block,
    // so main.zig:715 argmax only has to scan gridDim.x candidates afterwards
    float *part_val;
    int *part_idx;
    int row_offset;           // global index of row 0 (vocab shard offset)
    // Sharded runs, peer-write transport: the writer lane also stores every output value as an LL
    // word {value, epoch} straight into the peers' landing slots (p2p.hip), so the values travel
    // while the rest of the launch still runs and the gather that follows only has to collect.
    // Vector kernels only (launch_matvec reports whether it was honoured).  Device memory; may be null.
    const P2pArgs *push;
    const int *push_ctl;      // l2z_comm::d_ctl by value (saves a dependent load)
    int push_gi;              // index of the gather the outputs belong to
    // x is a gathered vector that is read as LL words from this rank's landing slot (xin.slots != null)
    LLIn xin;
};

// main.zig:361-389: scores, softmax, att.V for the local heads of one layer
struct AttnArgs {
    const float *q;        // (n_heads_local * head_size)
    const float *kcache;   // this layer: [kv_heads_local][seq_len][head_size] (head-major)
    const float *vcache;
    float *xb;             // (n_heads_local * head_size)
    const int *pos_ptr;
    int head_size;
    int kv_row;            // floats between consecutive timesteps of one kv head (head-major: head_size)
    size_t kv_head;        // floats between kv heads (head-major: seq_len * head_size)
    int kv_mul;
    int seq_len;
    const P2pArgs *push;   // as in MatvecArgs, for xb (fast / split-combine kernels); may be null
    const int *push_ctl;
    int push_gi;
    int tl_seq;            // attention launch number since the runstate was made (measurement builds only: L2Z_TIMELINE)
};


// fused_small.hip: rmsnorm + q/k/v rows + RoPE + KV write + attention of one head per block
// (main.zig:305-389) for small cache-resident MHA models; one launch instead of two
struct FusedQkvAttnArgs {
    const float *wq, *wk, *wv;   // this layer: (dim, dim) each (MHA: kv_dim == dim)
    const float *rms_w, *x;      // (dim)
    float *q_out;                // RunState.q (dim)
    float *kcache, *vcache;      // this layer: [heads][seq_len][head_size]; row pos of every head is written
    float *xb;                   // (dim) attention output
    const int *pos_ptr;
    const float2 *rope;
    int n;                       // dim
    int head_size, seq_len, kv_dim;
    int kv_row;                  // floats between timesteps of one head, and between heads (as AttnArgs)
    size_t kv_head;
};
bool fused_qkv_attn_supported(int dim, int n_heads, int n_kv_heads, int seq_len, int n_cus);
hipError_t launch_fused_qkv_attn(const FusedQkvAttnArgs &a, int n_heads, hipStream_t st);

// Launchers (matvec.hip, attention.hip, misc_kernels.hip).  All return a hipError_t from the launch.
// pushed: set to whether a.push was honoured (row kernel only)
// (a.push / a.xin are all-or-nothing: matvec_ll_supported tells beforehand whether they will be)
hipError_t launch_matvec(const MatvecArgs &a, int pro, int epi, int max_blocks_per_cu, int n_cus,
                         hipStream_t st, int *out_grid = nullptr, bool *pushed = nullptr);
// true if a launch with this width takes the vector kernels, which honour push and xin
bool matvec_ll_supported(int n);
// true if launch_attention / launch_attention_split will honour a.push (vector kernels only)
bool attention_push_supported(const AttnArgs &a);
int matvec_max_grid(int n_cus);
bool matvec_vector_width(int n);
// out: >= 8 * n_cus floats of scratch (never written in practice)
hipError_t launch_stream_read(const float *p, size_t n_floats, float *out, int n_cus, hipStream_t st);  // upper bound of the grid launch_matvec picks
hipError_t launch_attention(const AttnArgs &a, int n_heads_local, hipStream_t st, int form = 0);
// positions below this take the 256-thread speculative one-block-per-head form whatever seq_len is (attention.hip)
int attention_short_pos(int head_size, int seq_len);
// flash-decoding form: `nch` blocks per head, the last arriver combines (attention.hip)
int attention_split_chunks(int n_heads_local, int n_cus);
size_t attention_split_part_floats(int n_heads_local, int head_size, int nch);
bool attention_split_supported(const AttnArgs &a);
// arrivals: one int per local head, zero before the first launch (the kernel leaves them at zero)
// small: 256 threads per block instead of 1024 (positions below attention_split_wide_pos)
hipError_t launch_attention_split(const AttnArgs &a, int n_heads_local, int nch, float *part,
                                  int *arrivals, hipStream_t st, bool small = false);
int attention_split_wide_pos(int seq_len);
hipError_t launch_argmax(const ArgmaxArgs &a, hipStream_t st);
hipError_t launch_set_state(int token, int pos, int *token_ptr, int *pos_ptr, const float *tok_emb,
                            float *x, int dim, hipStream_t st);
hipError_t launch_rmsnorm(float *o, const float *x, const float *w, int n, hipStream_t st);
hipError_t launch_softmax(float *x, int n, hipStream_t st);
hipError_t launch_probs(float *probs, const float *logits, int n, float temperature, hipStream_t st);
hipError_t launch_dot(float *out, const float *x, const float *y, int n, hipStream_t st);
hipError_t launch_weighted_sum_rows(float *xout, int xout_len, const float *rows, int row_stride,
                                    const float *weights, int n_weights, hipStream_t st);
// row_len != 0: element i goes to dst[(i / row_len) * row_pitch + i % row_len] (rows of a strided matrix); idx_pitch != 0:
// its blob index is base_idx + (i / row_len) * idx_pitch + i % row_len (a column range of rows idx_pitch wide)
hipError_t launch_synth_fill(float *dst, uint64_t base_idx, uint64_t count, uint64_t seed,
                             float scale, float bias, hipStream_t st, uint64_t row_len = 0, uint64_t row_pitch = 0,
                             uint64_t idx_pitch = 0);
// out[i] = parts[0][i] + parts[1][i] + ... in that order (the emulated ranks' all-reduce of scheme B)
hipError_t launch_sum_parts(float *out, const float *const *parts, int n_parts, int n, hipStream_t st);
// dst row r (dpitch floats apart) = src row r (contiguous rows of cols floats); both on the device
hipError_t launch_copy_rows(float *dst, size_t dpitch, const float *src, size_t rows, size_t cols, hipStream_t st);
size_t attention_lds_bytes(int head_size, int seq_len, bool vec);

// ---- batched prefill (prefill_gemm.hip, prefill_skinny.hip, prefill_attention.hip) ----
// workspace of the tile GEMM's split-K family (prefill_gemm.hip SPLIT), owned by the runstate: accumulator
// dumps of the K ranges and one arrival counter per output tile (zero between launches)
struct SplitKWs {
    float *part;
    int *cnt;
    size_t part_floats;
    int cnt_ints;
    // the tile GEMM on the bf16 matrix cores: the launch's activation matrix as three planes of bf16 terms
    // (prefill_gemm.hip launch_split3), [chunk tokens][3][widest padded row] bf16; reallocated with the chunk scratch
    void *x3;
    size_t x3_bytes;
    // a second planes matrix: the gated hidden rows, written by the W1 | W3 launch's epilogue WHILE that launch reads its
    // own planes from x3 (unsharded pass: the producers of an activation matrix write its planes beside it -- rmsnorm and the
    // attention output into x3, the SwiGLU epilogue into x3b -- and the consumer's split launch is gone)
    void *x3b;
    size_t x3b_bytes;
};
// A residual product (Wo, W2) of the stream form whose K ranges' sums were LEFT in the workspace (unsharded pass): the rmsnorm
// launch that reads the residual stream next adds them -- range 0, 1, ... in order, then the residual: the sums the launch's
// own hand-over forms -- writes x back and normalises it.  The producing launch ends at its loop: no write-through, no
// arrival counter, no wait for the slowest sibling, no second pass over the sums (7-13 us per launch by the blocks' clocks).
struct DeferredSum {
    const float *part;   // [tile][range][feat / 32 waves][tm * 16][64 lanes]: the MFMA accumulators as the blocks held them
    int sk, feat, tm;    // K ranges per tile, features per tile (128 / 192 / 256), token tiles of 32
    bool valid;
};
// launch_prefill_gemm*'s planes argument: the consumer's planes are not there yet (the launcher splits x into ws->x3), stand
// in ws->x3 (the launch before multiplied the same x, or x's producer wrote them), or stand in ws->x3b
enum { PLANES_SPLIT = 0, PLANES_READY = 1, PLANES_READY_B = 2 };
constexpr int kSplitKMaxTokens = 256;  // longest chunk the split-K family takes
constexpr int kPanelWsRows = 6 * kSplitKMaxTokens;  // rows of the widest launch the partial-product workspace holds (panel kernel: ranges x 16 tms)
// K ranges per output tile for a [P, K] x [n_whole, K]^T product (1: the unsplit family).  Part of the
// arithmetic, so a function of the chunk length and the WHOLE model's matrix only -- never of a rank's share.
int prefill_split_k(long long n_whole, int P, int K, bool pair);
enum PrefillGemmEpi { PG_STORE = 0, PG_RESID = 1, PG_ROPE = 2, PG_ROPE_CACHE = 3, PG_CACHE = 4,
                      PG_SWIGLU = 5 };  // out = silu(out) * (X W^T): the W3 product merged into W1's
// PG_RESID: out = res + X W^T (res == nullptr: in place, res = out, ldres = ldo)
hipError_t launch_prefill_gemm(int epi, const float *x, int ldx, const float *w, float *out, int ldo,
                               int P, int N, int K, int pos0, const float2 *rope, int head_size,
                               hipStream_t st, const float *res = nullptr, int ldres = 0,
                               int n_scale = 1,  // n_scale: ranks the rows are sharded over (kernel-form choices look at the whole matrix)
                               size_t kv_head_stride = 0,  // PG_*CACHE: out is a head-major cache (MatvecArgs::kv_head_stride)
                               int sk = 1, const SplitKWs *ws = nullptr,   // sk > 1: the split-K family (prefill_split_k)
                               int ldw = 0,   // floats between rows of w (0: K; W1 / W3 of the device blob: 2 K)
                               long long n_launch_whole = 0,   // rows of the whole model's launch this product is a part of (q, k, v
                                                               // launched apart: dim + 2 kv_dim; 0: N * n_scale) -- the stream form's K ranges
                               int planes_ready = PLANES_SPLIT,    // PLANES_*: whether x's planes stand already, and where
                               DeferredSum *defer = nullptr);      // PG_RESID: != null: the launch MAY leave its K ranges' sums to the next
                                                                   // rmsnorm launch (sets valid; the stream form with > 1 range does)
// the stream form of the planes kernel (prefill_gemm.hip): which products take it, and their K ranges
bool x3_applies(long long n_whole, int K);
bool x3_stream_shape(long long n_whole, int P, int K);
int x3_stream_sk(long long n_whole, int P, int K);
hipError_t launch_prefill_gemm_qkv(const float *x, int ldx, const float *wq, const float *wk, const float *wv,
                                   float *q_out, int ldq, float *kcache, float *vcache, int ldkv, int P, int nq,
                                   int nkv, int K, int pos0, const float2 *rope, int head_size, hipStream_t st,
                                   size_t kv_head_stride = 0, int n_scale = 1, int sk = 1, const SplitKWs *ws = nullptr,
                                   int planes_ready = PLANES_SPLIT);
// planes_out: the launch may ALSO leave the planes of its output (K' = N columns, kp_out >= N bf16 per plane row) in
// ws->x3b; *planes_written says whether the form that ran did (the stream form does)
hipError_t launch_prefill_gemm_swiglu_pair(const float *x, int ldx, const float *w1, const float *w3,
                                           float *out, int ldo, int P, int N, int K, hipStream_t st,
                                           int n_scale = 1, int sk = 1, const SplitKWs *ws = nullptr, int ldw = 0,
                                           int planes_ready = PLANES_SPLIT, int kp_out = 0, bool *planes_written = nullptr);
hipError_t launch_prefill_gemm_kv_pair(const float *x, int ldx, const float *wk, const float *wv, float *kcache,
                                       float *vcache, int ldkv, int P, int nkv, int K, int pos0, const float2 *rope,
                                       int head_size, hipStream_t st, int n_scale = 1, size_t kv_head_stride = 0,
                                       int sk = 1, long long n_launch_whole = 0);
hipError_t launch_prefill_rmsnorm(float *o, int ldo, const float *x, const float *w, int n, int P,
                                  hipStream_t st,    // o: rows of ldo floats (the pad columns are left alone)
                                  void *x3 = nullptr, int kp = 0,    // != null: o's planes of bf16 terms too (x3[token][3][kp], kp == n)
                                  const DeferredSum *pending = nullptr);   // valid: x += the ranges' sums first (x is written back)
hipError_t launch_prefill_embed(float *x, const float *tok_emb, const int *tokens, int dim, int P,
                                hipStream_t st);
// kv_row / kv_head: floats between timesteps of one kv head / between kv heads (AttnArgs)
hipError_t launch_prefill_attention(const float *q, int ldq, const float *kcache, const float *vcache,
                                    float *out, int ldo, int pos0, int P, int n_heads, int head_size,
                                    int kv_row, size_t kv_head, int kv_mul, int seq_len, hipStream_t st,
                                    int n_heads_model = 0,   // heads of the whole model when n_heads is a shard's
                                    int form = 0,            // tests: 1 block per (head, query), 2 tiled, 3 flash; 0 by shape
                                    void *x3 = nullptr, int kp = 0,   // != null: out's planes of bf16 terms too (x3[token][3][kp]) ...
                                    bool *planes_written = nullptr);  // ... if the form that ran writes them (the flash form does)
// ---- prefill_panel.hip: chunks of <= 32 tokens of matrices that stream from HBM (K ranges with a resident X panel) ----
enum PanelEpi { PANEL_STORE = 0, PANEL_RESID = 1, PANEL_SWIGLU = 2, PANEL_QKV = 3 };
struct PanelProduct {
    const float *x;      // [P, K]
    int ldx;
    const float *w0, *w1, *w2;   // up to three [rows, K] matrices sharing x (unused: null, 0 rows); W1 | W3 interleaved = one matrix
    int rows0, rows1, rows2;
    int P, K;
    int mode;            // PanelEpi
    float *out;          // STORE / RESID: [P, ldo]; SWIGLU: [P, ldo], rows / 2 gated values; QKV: q [P, ldo]
    int ldo;
    const float *res;    // RESID: out = res + product
    int ldres;
    float *outk, *outv;  // QKV: the layer's key / value caches (rows1 == rows2 features each)
    int ldkv, head_size, pos0;
    size_t kv_head_stride;
    const float2 *rope;
};
// a function of the WHOLE model's matrix (n_whole rows) and the chunk length: a rank's shard takes what the unsharded pass takes
// (widest_whole: rows of the whole model's widest launch + 128, what sizes the split-K / partial-product workspace)
bool prefill_panel_shape(long long n_whole, int P, int K, long long widest_whole);
int prefill_panel_max_tokens();
// The next chunk of THIS model's batched prompt pass: the planner's (tunables.cpp prefill_next_chunk), except that a tail
// past a step of the tile GEMM's cost staircase is cut into the step's worth + the rest on a model whose matrices take the
// short-chunk kernels: 129 ... 224 tokens = 128 + rest and 257 ... 384 = 256 + rest on the bf16 cores (129 ... 160 and
// 257 ... 288 on the f32 cores), and 673 ... 1023 tokens are ONE chunk on the bf16 cores.
// A function of the WHOLE model's shape and the tokens left: every rank of a shard group cuts the same chunks.
int prefill_next_chunk_of(const l2z_config &c, int remaining);
// hipErrorNotSupported: this rank's rows / pointers / workspace do not take the kernel (rows % 16, alignment) -- callers that
// asked prefill_panel_shape first treat that as an error on a shard (the unsharded pass would have taken it)
hipError_t launch_prefill_panel(const PanelProduct &p, int n_cus, const SplitKWs *ws, hipStream_t st);
int prefill_tile_form(int N, int P, int pair);  // 0: 128x64, 1: 64x64, 2: 32x64, 3: 32x32, 4: 128x128
size_t matvec_lds_bytes(int n);

}  // namespace l2z
