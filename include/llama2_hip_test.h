/*
 * llama2_hip_test.h -- entry points of libllama2_hip.so that exist for TESTS and MEASUREMENT only.
 *
 * Nothing here is part of the drop-in boundary (include/llama2_hip.h): a host that replaces
 * src/main.zig's transformer() never calls these.  tests/, bench.py and scripts/ do:
 *   - kernel-level hooks named after the reference's math functions (src/main.zig:432-726), so the
 *     reference's own unit-test vectors (main.zig:1078-1150) can be run against the device code;
 *   - l2z_attention_decode: the decode attention kernels the forward pass launches, driven directly;
 *   - read-back of device state, the seeded synthetic-checkpoint generator, emulated ranks;
 *   - per-kernel timing and the streaming-read probe behind bench.py's roofline;
 *   - l2z_option_set: the tuning knobs of csrc/tunables.h from inside a process.
 */
#ifndef LLAMA2_HIP_TEST_H
#define LLAMA2_HIP_TEST_H

#include "llama2_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- synthetic checkpoints / state read-back ---- */
/* Same layout, filled ON DEVICE by the seeded generator of DESIGN.md
 * "Synthetic checkpoints" (no checkpoint exists in the build image, and a 27 GB
 * PCIe upload is not part of the measured path). */
int l2z_weights_init_synthetic(const l2z_config *config, int shared_weights, uint64_t seed,
                               const l2z_comm *comm, l2z_weights **out);
/* Copy `count` floats starting at index `offset` of the checkpoint's weight blob (FILE order, main.zig:85-112)
 * back to the host -- the device copy keeps W1 | W3 row-interleaved, this call undoes that
 * (single-GPU weights only; used by tests to check uploads / the generator). */
int l2z_weights_read(const l2z_weights *w, size_t offset, size_t count, float *out);

/* copy a named RunState buffer to the host: "x","xb","hb","q","att","logits",
 * "key_cache","value_cache" (tests only) */
int l2z_runstate_read(l2z_runstate *s, const char *name, size_t offset, size_t count, float *out);


/* ---- measurement support ----
 * l2z_stream_read_probe streams `slice_bytes` pieces of the resident weight blob (0 = all of it)
 * through a pure read kernel `reps` times, a different piece per launch, and returns the average
 * and best read rate in GB/s: the measured ceiling bench.py quotes beside the 8 TB/s HBM3E spec
 * (SURVEY.md 8d "also report against a measured ... on the same box").
 * l2z_profile_forward runs ONE forward pass (+ argmax/hand-over) eagerly with
 * a HIP event pair around every kernel launch, recorded on the runstate's own
 * stream, and returns per-kind total device time (ms) and launch count.
 * Kinds, in slot order (l2z_kind_name): 0 "qkv", 1 "attn", 2 "wo", 3 "ffn13",
 * 4 "ffn2", 5 "cls", 6 "argmax" (0 launches when the classifier's last block hands the loop over
 * itself), 7 "gather" (sharded runs: gather launches); n_kinds must be >= 8.  bench.py derives the
 * roofline of the dominant kernel from this, in situ: every layer streams
 * its own weights, so nothing is re-read from cache between launches.  The
 * numbers must agree with rocprofv3 --kernel-trace --stats (profiles/). */
#define L2Z_N_KINDS 8
int l2z_stream_read_probe(l2z_runstate *s, const l2z_weights *w, size_t slice_bytes, int reps,
                          double *avg_gbps, double *best_gbps);
/* the same pieces copied device to device (hipMemcpyAsync) into a scratch allocation: GB/s of bytes COPIED
 * (memory traffic is twice that), SURVEY.md 8d's "measured device-to-device copy on the same box" */
int l2z_d2d_copy_probe(l2z_runstate *s, const l2z_weights *w, size_t slice_bytes, int reps,
                       double *avg_gbps, double *best_gbps);
int l2z_profile_forward(int token, int pos, const l2z_config *config, l2z_runstate *s,
                        const l2z_weights *w, double *ms_by_kind, int *launches_by_kind,
                        int n_kinds);
int l2z_kind_name(int kind, char *out, size_t cap);
/* Average duration of ONE launch of `kind` (0..6) at position `pos`: `reps` passes of that kind's
 * launches for every layer, back to back between ONE event pair on the runstate's stream -- no
 * per-launch event overhead, every launch streams its own layer's weights.  Comparable with
 * rocprofv3 --kernel-trace durations.  Unsharded runstates only. */
int l2z_time_kind(int kind, int pos, const l2z_config *config, l2z_runstate *s, const l2z_weights *w,
                  int reps, double *avg_ms_per_launch, int *launches);

/* ---- kernel-level test hooks (host pointers in and out; same device code
 *      the forward pass runs).  Names follow src/main.zig. ---- */
int l2z_matmul(float *xout, const float *x, const float *w, size_t n, size_t d);      /* :485 */
int l2z_matmul_fused(int N, float *const *outs, const float *x, const float *const *ws, size_t n,
                     size_t d);                                                        /* :530 */
int l2z_rmsnorm(float *o, const float *x, const float *w, size_t n);                   /* :432 */
int l2z_softmax(float *x, size_t n);                                                   /* :687 */
int l2z_vector_dot_product(float *out, const float *x, const float *y, size_t n);      /* :503 */
int l2z_vector_weighted_sum_rows(float *xout, size_t xout_len, const float *rows, size_t rows_len,
                                 size_t row_stride, const float *weights, size_t n_weights); /* :657 */
int l2z_argmax_host(const float *x, size_t n, size_t *out_index);                      /* :715 */

/* The decode attention of ONE layer (src/main.zig:361-389: scores :367-375, softmax :378 / :687-706,
 * weighted sum of V rows :381-388 / :657-685) through the kernels the forward pass launches.
 *   q [n_heads*head_size]; kcache, vcache [seq_len * kv_dim] (rows 0..pos are read);
 *   out [n_heads*head_size].   form: 0 = what the forward pass would pick at `pos`,
 *   1 = one block per head, 256 threads (speculative first round: seq_len <= 512 models),
 *   2 = one block per head, 1024 threads, 3 = split over `nch` blocks per head + combine
 *   (nch 0: the runstate default for n_heads), 4 = generic kernel. */
int l2z_attention_decode(int form, int nch, float *out, const float *q, const float *kcache,
                         const float *vcache, int pos, int n_heads, int n_kv_heads, int head_size,
                         int seq_len);

/* The batched prefill's attention kernels (llama2.zig_amd/csrc/prefill_attention.hip) for the n_queries
 * queries at positions pos0 .. pos0 + n_queries - 1 of one layer: q and out are [n_queries][n_heads * head_size],
 * the caches [seq_len][n_kv_heads * head_size] with rows 0 .. pos0 + n_queries - 1 filled.
 * form: 0 as l2z_prefill picks, 1 block per (head, query), 2 tiled with the softmax in LDS, 3 / 4 flash form
 * with one / two key parts (head sizes 64 and 128). */
int l2z_prefill_attention(int form, float *out, const float *q, const float *kcache, const float *vcache,
                          int pos0, int n_queries, int n_heads, int n_kv_heads, int head_size, int seq_len);

/* Host-side shard geometry (no device needed): out[0..9] = dim0, dim_loc, kvd_loc, heads_loc, hid0, hid_loc, v0, v_loc of
 * rank `rank` of `world` (scheme A: rows / heads owned), then dimc_pad, hidc_pad: the zero-padded row width of its Wo / W2
 * column shards under scheme B (multiples of 256 floats above 768, of 4 below).  L2Z_ERR_INVALID when the shape does not
 * split over `world` ranks. */
int l2z_shard_plan(const l2z_config *config, int rank, int world, int *out, int cap);

/* Host-side planning of the batched prefill (no device needed).  l2z_prefill_plan: the chunk lengths a
 * prompt of n_tokens is cut into (returns their number, writes up to cap of them).  l2z_prefill_tile: the
 * output tile of the direct-to-LDS GEMM for an [n_tokens, n_features] product -- 0: 128x64, 1: 64x64,
 * 2: 32x64, 3: 32x32, 4: 128x128 (all forms give the same bits; the choice fills the CUs). */
int l2z_prefill_plan(int n_tokens, int *chunks, int cap);
/* The same for a given model: where the model's matrices take the K-range panel kernel (chunks of 17 ... 64 tokens of
 * matrices that stream from HBM) a tail of 65 ... 96 tokens is cut in two chunks of that range (48 | 64 tokens first),
 * and a tail of 129 ... 160 / 257 ... 288 tokens into 128 / 256 + the rest. */
int l2z_prefill_plan_model(const l2z_config *config, int n_tokens, int *chunks, int cap);
int l2z_prefill_tile(int n_features, int n_tokens, int paired);
/* K ranges per output tile of the tile GEMM's split-K family for an [n_tokens, k] x [n_features_whole, k]^T
 * product (1: the unsplit family; > 1 also means the tile kernel instead of the short-prompt kernels).  Part of
 * the arithmetic -- the range partials are added in range order -- hence a function of the chunk length and the
 * WHOLE model's matrix, never of a rank's share of its rows. */
int l2z_prefill_split_k(long long n_features_whole, int n_tokens, int k, int paired);
/* The matrix cores of an [n_tokens, k] x [n_features_whole, k]^T product (round 6): 0 the f32 ones
 * (v_mfma_f32_32x32x2_f32, an fmaf chain), 1 the bf16 ones over three-term splits of both operands (six
 * v_mfma_f32_32x32x16_bf16 per 16 k; matrices that stream from HBM, or all with L2Z_PF_X3=2) in the tile forms, n >= 2: the
 * STREAM form of that kernel (chunks of L2Z_PF_X3_STREAM_MIN = 33 ... 128 tokens) with n - 1 K ranges per output tile -- part of
 * the arithmetic, a function of the chunk length and the WHOLE model's matrix.  k: the product's K (rounded up to 64 here). */
int l2z_prefill_cores(long long n_features_whole, int n_tokens, int k);

/* ---- emulated ranks ---- */
/* Testing support: N emulated ranks in ONE process on ONE GPU (RCCL refuses two ranks on
 * one device).  l2z_comm_init_emulated makes a rank descriptor without a communicator;
 * weights / runstates built with it hold exactly rank r's shard; l2z_emu_transformer runs
 * one forward pass for all ranks, interleaved stage by stage, doing each all-gather as
 * device-to-device copies.  Afterwards every rank's logits must equal the unsharded pass. */
int l2z_comm_init_emulated(int rank, int world, int device, l2z_comm **out);
int l2z_emu_transformer(int n_ranks, l2z_runstate *const *ss, const l2z_weights *const *ws,
                        int token, int pos);
/* l2z_prefill for the emulated ranks: every rank's own launches of each stage (llama2.zig_amd/csrc/
 * prefill_host.cpp), the [tokens, n / world] activation blocks exchanged as device-to-device copies.
 * Afterwards every rank's KV shard and logits must equal the unsharded l2z_prefill, bit for bit. */
int l2z_emu_prefill(int n_ranks, l2z_runstate *const *ss, const l2z_weights *const *ws,
                    const int32_t *tokens, int n_tokens, int pos0);

/* What a shard group's transports are (bench.py's comm{} record): the rank count RCCL itself reports
 * for the communicator (ncclCommCount; 0 when the group has none) and whether the peer-write arenas
 * are connected. */
int l2z_comm_transports(const l2z_comm *c, int *rccl_ranks, int *p2p_connected);

/* Measurement: after l2z_comm_p2p_export, connect this rank ALONE -- every peer's arena is a local sink, this rank's own zeroed slots satisfy every hand-over's
 * wait is satisfied by the zeroed slots -- so that ONE rank of an N-rank group runs its whole sharded pass (launches,
 * pushes, polls, gather / reduce launches) on a GPU by itself: the per-rank time with free hand-overs.  Results are
 * meaningless (the peers' slices read as zeros). */
int l2z_comm_p2p_connect_solo(l2z_comm *c);

/* Diagnostics of the peer-write transport, for bench.py's N > 1 legs (what a hand-over costs between two ranks of
 * THIS group on THIS box -- xGMI when they sit on two GPUs).  Both are made by two ranks at once, like every call of a
 * shard group: l2z_comm_p2p_pingpong by `other` and by this rank with opposite `initiator` flags and the same iters --
 * *rtt_us = microseconds per round trip of one 8-byte LL word each way (device clock, one polling thread per side);
 * l2z_comm_peer_copy_probe by one rank only: `bytes` into the other's arena by the runtime's device-to-device copy,
 * each copy synchronised: microseconds per copy (host clock). */
int l2z_comm_p2p_pingpong(l2z_comm *c, int other, int initiator, int iters, double *rtt_us);
int l2z_comm_peer_copy_probe(l2z_comm *c, int other, size_t bytes, int iters, double *us_per_copy);

/* Loads RCCL (dlopen) now and reports the file the process got and ncclGetVersion's code.  A process that imports
 * PyTorch afterwards keeps THIS copy (same SONAME); one that imported it before gets torch's bundled copy. */
int l2z_comm_rccl_info(char *path_out, size_t cap, int *version);

/* The structure a runstate runs: bit 3 = sharding scheme B (L2Z_SCHEME_B: column-sharded Wo / W2 + all-reduces);
 * 0 = the default.  (Bits 0-2 named round 4's opt-in decode forms, removed in round 5: always 0.)  Tests that ask
 * for an option check here that they got it. */
int l2z_runstate_form(const l2z_runstate *s, int *form);

/* Set one tuning knob by its environment-variable name (csrc/tunables.h), e.g. ("L2Z_P2P_CONSUME", 0).
 * Applies to objects created afterwards.  L2Z_ERR_INVALID for an unknown name. */
int l2z_option_set(const char *env_name, long long value);

#ifdef __cplusplus
}
#endif
#endif /* LLAMA2_HIP_TEST_H */
