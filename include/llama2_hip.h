/*
 * llama2_hip.h -- C ABI of the MI355X (gfx950) forward pass for cgbur/llama2.zig.
 *
 * The reference has no FFI today: the boundary is the Zig-internal call
 *     transformer(token, pos, *Config, *RunState, *Weights) void     src/main.zig:285
 * with its sole call site at src/main.zig:996.  This header gives that call,
 * and the three objects it takes, a C ABI with the same names, argument
 * meaning and ownership, so a maintainer replaces each Zig call by the
 * matching extern (INTEGRATION.md shows the Zig `extern fn` block).
 *
 *   reference (src/main.zig)                 this library
 *   ---------------------------------------  -----------------------------------
 *   ConfigReader / Config        :17-49      l2z_config (same 7 x i32 layout)
 *   Weights.init(config,data,shared) :73     l2z_weights_init      (uploads to HBM)
 *   RunState.init(alloc,config)  :137        l2z_runstate_init     (device buffers)
 *   RunState.deinit              :156        l2z_runstate_free
 *   transformer(token,pos,c,s,w) :285        l2z_transformer
 *   argmax(state.logits)         :715,:1003  l2z_argmax            (on device)
 *   state.logits after return    :1005-1012  l2z_logits_read       (D2H, for samplers)
 *   logits / temperature, softmax :1005-1008  l2z_probs_read        (on the device, then D2H)
 *   while (pos < seq_len) loop at -t 0 :995  l2z_greedy_begin / l2z_greedy_run
 *   matmul, rmsnorm, softmax, ... :432-726   kernel-level hooks of the same names, for tests only:
 *                                            include/llama2_hip_test.h
 *
 * Ownership: the caller owns config; weights and runstate handles own their
 * device memory.  The host blob passed to l2z_weights_init may be freed (or
 * un-mmapped) as soon as the call returns.  One runstate = one sequence, used
 * from one thread at a time, pos strictly increasing from 0 -- exactly the
 * reference's contract (main.zig:994-995).
 *
 * Errors: the reference's transformer() cannot fail; a device path can.  Every
 * entry point returns L2Z_OK (0) or a negative l2z_status; l2z_last_error()
 * returns a thread-local message.  There is NO CPU fallback: without a gfx950
 * device every compute entry point fails with L2Z_ERR_NO_DEVICE.
 *
 * All floats are IEEE f32, exactly as in the checkpoint.
 */
#ifndef LLAMA2_HIP_H
#define LLAMA2_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define L2Z_ABI_VERSION 2

typedef enum l2z_status {
    L2Z_OK = 0,
    L2Z_ERR_INVALID = -1,    /* bad argument / shape the kernels do not support */
    L2Z_ERR_NO_DEVICE = -2,  /* no HIP device, or device is not usable */
    L2Z_ERR_HIP = -3,        /* a HIP runtime call failed */
    L2Z_ERR_OOM = -4,        /* device or host allocation failed */
    L2Z_ERR_COMM = -5,       /* RCCL failure (multi-GPU only) */
    L2Z_ERR_STATE = -6       /* call sequence violates the contract (e.g. pos out of range) */
} l2z_status;

/* src/main.zig:17-25 ConfigReader: 7 x i32, little endian, the first 28 bytes
 * of a checkpoint.  vocab_size here is already abs()'d (main.zig:944); the
 * sign is passed separately as `shared_weights` (main.zig:943). */
typedef struct l2z_config {
    int32_t dim;        /* transformer dimension */
    int32_t hidden_dim; /* ffn hidden dimension */
    int32_t n_layers;
    int32_t n_heads;
    int32_t n_kv_heads; /* <= n_heads, divides it (GQA) */
    int32_t vocab_size;
    int32_t seq_len;    /* max sequence length = KV-cache rows per layer */
} l2z_config;

typedef struct l2z_weights l2z_weights;   /* src/main.zig:53  Weights, resident in HBM */
typedef struct l2z_runstate l2z_runstate; /* src/main.zig:119 RunState, resident in HBM */
typedef struct l2z_comm l2z_comm;         /* multi-GPU shard group (no reference equivalent) */

/* ---- library / device ---- */
int l2z_abi_version(void);
const char *l2z_last_error(void);
int l2z_device_count(int *out_n);
/* name (<= cap bytes), CU count, HBM bytes of device `dev` */
int l2z_device_info(int dev, char *name, size_t cap, int *out_cus, uint64_t *out_hbm_bytes);

/* ---- Weights: src/main.zig:73 Weights.init(config, data, shared_weights) ----
 * `data` is the f32 blob that follows the 28-byte header, n_floats long, in
 * the Weights.init carve order (main.zig:85-112) including the unused
 * freq_cis region.  comm == NULL: the whole blob is uploaded as ONE device
 * allocation with the same layout.  comm != NULL: only this rank's rows of
 * every matrix are uploaded (heads / output rows, DESIGN.md "Sharding"). */
int l2z_weights_init(const l2z_config *config, const float *data, size_t n_floats,
                     int shared_weights, const l2z_comm *comm, l2z_weights **out);
void l2z_weights_free(l2z_weights *w);

/* ---- RunState: src/main.zig:137 RunState.init / :156 deinit ----
 * Allocates x, xb, hb, q, att, logits and the (layer, seq_len, kv_dim) key and
 * value caches in HBM, plus the RoPE cos/sin table (seq_len, head_size/2) that
 * replaces the per-layer pow/cos/sin of main.zig:338-342. */
int l2z_runstate_init(const l2z_config *config, const l2z_comm *comm, l2z_runstate **out);
void l2z_runstate_free(l2z_runstate *s);

/* ---- src/main.zig:285 transformer(token, pos, config, s, w) ----
 * One decoder forward pass; on return the logits for `pos` are in the
 * runstate (device) and the KV-cache rows `pos` are written in every layer.
 * Asynchronous on the runstate's stream; l2z_argmax / l2z_logits_read sync. */
int l2z_transformer(int token, int pos, const l2z_config *config, l2z_runstate *s,
                    const l2z_weights *w);
/* src/main.zig:715 argmax over s.logits, on device (strict '>' : lowest index wins ties).
 * SHARDED runstates: l2z_argmax, l2z_logits_read and l2z_probs_read are COLLECTIVE after greedy steps
 * (l2z_greedy_run) on the peer-write transport -- those steps exchange one argmax candidate per rank instead of
 * gathering the logits, so the first call that needs the whole vector gathers it, and every rank of the group
 * must make that call (as every rank makes every other call); a rank that calls alone waits L2Z_P2P_TIMEOUT_S
 * and gets L2Z_ERR_COMM.  After l2z_transformer / l2z_prefill the logits are already whole: plain reads. */
int l2z_argmax(l2z_runstate *s, int *out_token);
/* copy s.logits (vocab_size floats) to the host (sharded: see l2z_argmax) */
int l2z_logits_read(l2z_runstate *s, float *out_logits);
/* src/main.zig:1005-1008 on the device: out_probs[i] = softmax(logits / temperature)[i] (temperature > 0),
 * then the device-to-host copy of vocab_size floats; the caller goes on with sample / sample_top_p
 * (:1009-1012).  32000 exp() on one host core take longer than a small model's forward pass. */
int l2z_probs_read(l2z_runstate *s, float temperature, float *out_probs);
/* ---- src/main.zig:987-1042, the generation loop at temperature 0 ----
 * Runs entirely on the device: the forward pass, argmax, the prompt override
 * (main.zig:999-1000) and the token/pos hand-over to the next step are one
 * hipGraph replayed per position, with no host round trip per token.
 *   l2z_greedy_begin : token = BOS(1), pos = 0, install the prompt
 *   l2z_greedy_run   : run `n_steps` more positions, write `next` of each to
 *                      out_tokens; stops early after a BOS (main.zig:1017) and
 *                      at seq_len; *out_n = positions actually produced.
 */
int l2z_greedy_begin(l2z_runstate *s, const int32_t *prompt, int n_prompt);
int l2z_greedy_run(const l2z_config *config, l2z_runstate *s, const l2z_weights *w, int n_steps,
                   int32_t *out_tokens, int *out_n);

/* ---- batched prefill (no reference equivalent: src/main.zig:999-1000
 * feeds the prompt one token at a time) ----
 * Same state change as l2z_transformer(tokens[i], pos0 + i) for i = 0 .. n_tokens-1 -- the
 * KV-cache rows pos0 .. pos0+n_tokens-1 of every layer are written and the logits of the LAST
 * position are left in the runstate (l2z_argmax / l2z_logits_read) -- but each weight matrix is
 * streamed once per chunk of up to 1024 tokens and multiplied as a dense GEMM on the matrix cores:
 * the fp32 ones (v_mfma_f32_32x32x2_f32) for models whose matrices stay in the caches; for matrices
 * that stream from HBM the bf16 ones, f32-ACCURATELY -- both operands cut into three bf16 terms
 * (exact splits), the six products of order >= 2^-16 summed in f32 (error against float64 not above
 * the fp32 cores' own chain; L2Z_PF_X3=0 in the environment keeps the fp32 cores everywhere).  Values
 * agree with the token-by-token path up to summation order.  Dims must be multiples of 4 (else L2Z_ERR_INVALID, and the caller loops over
 * l2z_transformer).  On a sharded runstate every rank of the group makes the same call: the pass is
 * row-sharded like the decode pass and bit-identical to the unsharded one; it needs a transport for
 * [1024, hidden_dim] matrices -- the peer-write arena's bulk regions (allocated by
 * l2z_comm_p2p_export unless L2Z_P2P_BULK_MB=0) or an RCCL communicator -- else L2Z_ERR_INVALID.
 * l2z_greedy_run uses the same pass for the prompt positions when its first call after
 * l2z_greedy_begin asks for at least n_prompt steps, n_prompt >= L2Z_PREFILL_MIN_PROMPT and no
 * prompt token is BOS; L2Z_PREFILL=0 in the environment keeps the stepped loop. */
#define L2Z_PREFILL_MIN_PROMPT 4
int l2z_prefill(const int32_t *tokens, int n_tokens, int pos0, const l2z_config *config,
                l2z_runstate *s, const l2z_weights *w);

/* wait for everything queued on the runstate's stream */
int l2z_synchronize(l2z_runstate *s);

/* ---- multi-GPU shard group: one process per GPU, xGMI ----
 * The reference is single-threaded and single-device; this is what the build
 * adds.  Two transports for the per-layer all-gathers:
 *  - RCCL: id is an opaque 128-byte ncclUniqueId made on rank 0 and distributed by the launcher
 *    (bench.py uses torch.distributed/gloo for that);
 *  - peer writes: l2z_comm_init(rank, world, NULL, device, &c), then every rank exports the IPC
 *    handle of its landing arena (l2z_comm_p2p_export), the launcher all-gathers the 64-byte
 *    handles in rank order, and l2z_comm_p2p_connect maps the peers.  A gather is then one small
 *    kernel of direct 8-byte {value, epoch} stores into the peers' memory, polled by the receiver --
 *    no fences, no collective library, and it can be captured in the step graph.  Preferred when both are set up (L2Z_COMM=rccl overrides).
 *    max_vector_floats must be >= max(dim, hidden_dim, vocab_size) of every config used with the
 *    group: l2z_runstate_init refuses (L2Z_ERR_COMM) a config the landing slots cannot hold.
 *    The arena also carries two bulk regions of max_vector_floats x 1024 floats (L2Z_P2P_BULK_MB
 *    overrides, 0 = none) for the sharded prefill's activation matrices: plain 16-byte peer stores
 *    and one flag per sender.
 */
#define L2Z_COMM_ID_BYTES 128
int l2z_comm_unique_id(void *out_id);
int l2z_comm_init(int rank, int world, const void *id, int device, l2z_comm **out);
#define L2Z_COMM_IPC_BYTES 64
/* max_vector_floats: the longest vector that will be gathered = max(dim, hidden_dim, vocab_size) */
int l2z_comm_p2p_export(l2z_comm *c, size_t max_vector_floats, void *handle_out);
/* the same with the bulk regions sized separately: max_matrix_width = max(dim, hidden_dim), the widest
 * [tokens, n] matrix the sharded prefill gathers (the vocabulary only ever travels as a vector), so the
 * arena holds 2 x max_matrix_width x 1024 floats of bulk space instead of 2 x max_vector_floats x 1024 */
int l2z_comm_p2p_export_sized(l2z_comm *c, size_t max_vector_floats, size_t max_matrix_width, void *handle_out);
int l2z_comm_p2p_connect(l2z_comm *c, const void *handles /* world x L2Z_COMM_IPC_BYTES */);
int l2z_comm_rank(const l2z_comm *c, int *rank, int *world);
void l2z_comm_free(l2z_comm *c);
/* Pure host logic, no GPU needed: the row range [*r0,*r1) of a `rows`-row
 * tensor owned by `rank` of `world`, in units of `granule` rows (head_size for
 * q/k/v so shards are whole heads, 1 otherwise).  Fails if not divisible. */
int l2z_shard_range(int64_t rows, int64_t granule, int rank, int world, int64_t *r0, int64_t *r1);

#ifdef __cplusplus
}
#endif
#endif /* LLAMA2_HIP_H */
