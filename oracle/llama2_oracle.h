/*
 * llama2_oracle.h -- CPU restatement of cgbur/llama2.zig's forward pass.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may link or call it, and only
 * as the checker.  The product path (llama2.zig_amd/csrc) never includes this
 * header and never falls back to it.
 *
 * PARITY STATUS: the per-kernel functions are pinned against the reference's
 * own known-answer tests (src/main.zig:1078-1150, see tests/golden/).  The
 * whole-pass function orc_transformer() is "parity unpinned": the reference
 * holds no golden vector for transformer(), there is no Zig 0.16 compiler and
 * no checkpoint in this image, so it can only be checked against independent
 * implementations of the same architecture: a float64 numpy restatement
 * (tests/test_oracle_cpu.py) and Hugging Face's LlamaForCausalLM on the same
 * seeded checkpoint (tests/test_oracle_vs_hf.py: logits equal to <= 1e-5 at
 * every position) -- neither of which is the Zig binary.
 *
 * Every function cites the reference lines (src/main.zig) it restates.
 * The reference's arithmetic depends on the host through
 * DEFAULT_VECTOR_WIDTH (main.zig:7) and on LLVM's freedom under
 * @setFloatMode(.optimized) (main.zig:11-13), so three knobs select which
 * reading of the reference is emulated (orc_set_mode).
 */
#ifndef LLAMA2_ORACLE_H
#define LLAMA2_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* main.zig:17-25 ConfigReader (7 x i32 LE) / main.zig:41-49 Config */
typedef struct {
    int32_t dim;
    int32_t hidden_dim;
    int32_t n_layers;
    int32_t n_heads;
    int32_t n_kv_heads;
    int32_t vocab_size; /* already abs()'d, main.zig:944 */
    int32_t seq_len;
} orc_config;

/* main.zig:53-72 Weights: pointers into one flat f32 blob */
typedef struct {
    const float *token_embedding_table; /* (vocab_size, dim) */
    const float *rms_att_weight;        /* (layer, dim) */
    const float *rms_ffn_weight;        /* (layer, dim) */
    const float *wq;                    /* (layer, dim, dim) */
    const float *wk;                    /* (layer, kv_dim, dim) */
    const float *wv;                    /* (layer, kv_dim, dim) */
    const float *wo;                    /* (layer, dim, dim) */
    const float *w1;                    /* (layer, hidden_dim, dim) */
    const float *w2;                    /* (layer, dim, hidden_dim) */
    const float *w3;                    /* (layer, hidden_dim, dim) */
    const float *rms_final_weight;      /* (dim,) */
    const float *freq_cis_real;         /* carved, never read (main.zig:68) */
    const float *freq_cis_imag;
    const float *wcls;                  /* (vocab_size, dim) */
} orc_weights;

/* main.zig:119-135 RunState (logits_indexed is sampler-only, omitted) */
typedef struct {
    float *x, *xb, *xb2, *hb, *hb2, *q, *k, *v, *att, *logits;
    float *key_cache, *value_cache; /* (layer, seq_len, kv_dim) */
} orc_runstate;

/*
 * Which reading of the reference to emulate.
 *   vector_width : DEFAULT_VECTOR_WIDTH, 4 | 8 | 16  (main.zig:7; 8 on AVX2
 *                  hosts such as the README's Ryzen 5900X, 16 on AVX-512)
 *   use_fma      : 0 = separate mul+add (strict), 1 = fused (what LLVM may do
 *                  under .optimized float mode on an FMA host)
 *   tree_reduce  : 0 = @reduce(.Add) sequential lane 0..VW-1 (strict),
 *                  1 = log2 halving tree (LLVM's reassociated lowering)
 * Default after load: (8, 0, 0).
 */
void orc_set_mode(int vector_width, int use_fma, int tree_reduce);
void orc_get_mode(int *vector_width, int *use_fma, int *tree_reduce);

/* ---- math kernels, main.zig:432-726 ---- */
void orc_rmsnorm(float *o, const float *x, const float *w, size_t n);           /* :432-468 */
void orc_matmul(float *xout, const float *x, const float *w, size_t n, size_t d); /* :485-498 */
/* :530-605, N in {1,2,3}; outs[j] has d rows, ws[j] is (d,n) row-major */
void orc_matmul_fused(int N, float *const *outs, const float *x, const float *const *ws,
                      size_t n, size_t d);
float orc_vector_dot_product(const float *x, const float *y, size_t n);          /* :503-527 */
void orc_vector_mul(float *x, const float *y, size_t n);                          /* :608-628 */
void orc_vector_weighted_sum(float *xout, const float *x, float y, size_t n);     /* :632-653 (dead code in ref) */
void orc_vector_weighted_sum_rows(float *xout, size_t xout_len, const float *rows,
                                  size_t row_stride, const float *weights, size_t n_weights); /* :657-685 */
void orc_softmax(float *x, size_t n);                                             /* :687-706 */
void orc_accum(float *a, const float *b, size_t n);                               /* :708-713 */
size_t orc_argmax(const float *x, size_t n);                                      /* :715-726 */
/* the inline RoPE of transformer(), :336-351, in place on q [dim] and k [kv_dim] */
void orc_rope(float *q, float *k, size_t pos, size_t dim, size_t kv_dim, size_t head_size);

/* ---- checkpoint / state, main.zig:73-115, :137-154 ---- */
/* number of f32 in the weight blob (after the 28-byte header) */
size_t orc_weights_count(const orc_config *c, int shared_weights);
void orc_weights_init(orc_weights *w, const orc_config *c, const float *data, int shared_weights);
int orc_runstate_init(orc_runstate *s, const orc_config *c); /* 0 ok, -1 oom */
void orc_runstate_free(orc_runstate *s);

/* ---- the forward pass, main.zig:285-430 ---- */
void orc_transformer(size_t token, size_t pos, const orc_config *c, orc_runstate *s,
                     const orc_weights *w);

/*
 * Greedy generation loop, main.zig:987-1042 at temperature 0: start from
 * BOS=1, feed prompt[pos] while pos < n_prompt, otherwise argmax; stop on
 * next==1 or after `steps` positions.  Writes `next` for every executed
 * position to out_tokens (including the terminating BOS if hit) and, if
 * margins != NULL, the top-1 minus top-2 logit gap per position (tie-margin
 * report).  Returns the number of positions executed.
 */
size_t orc_generate_greedy(const orc_config *c, orc_runstate *s, const orc_weights *w,
                           const int32_t *prompt, size_t n_prompt, size_t steps,
                           int32_t *out_tokens, float *margins);

/* ---- seeded synthetic checkpoints (no real .bin exists in the image) ----
 * value(idx) = bias + scale * r(idx, seed), r uniform on a 2^-22 grid in
 * [-1, 1); idx is the flat f32 index in the blob.  The same generator is
 * implemented on the device (csrc/misc_kernels.hip: synth_fill_kernel) and in numpy (checkpoint.py);
 * tests check all three agree bit-for-bit.
 */
float orc_synth_value(uint64_t idx, uint64_t seed, float scale, float bias);
/* fill the whole blob for a config with per-tensor (scale,bias) as documented
 * in DESIGN.md "Synthetic checkpoints"; n_threads<=1 runs serially */
void orc_synth_fill(float *data, const orc_config *c, int shared_weights, uint64_t seed,
                    int n_threads);
/* fill rows [row0,row0+nrows) of one (d,n) tensor that starts at blob index base */
void orc_synth_fill_range(float *dst, uint64_t base_idx, uint64_t count, uint64_t seed,
                          float scale, float bias);

#ifdef __cplusplus
}
#endif
#endif
