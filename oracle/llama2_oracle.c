/*
 * llama2_oracle.c -- CPU restatement of cgbur/llama2.zig's forward pass.
 *
 * TEST INFRASTRUCTURE ONLY (see llama2_oracle.h): the checker that the HIP
 * path is compared against, and the "port" CPU baseline timed by bench.py.
 * Never linked into, loaded by, or used as a fallback for the product library.
 *
 * PARITY: kernels pinned by the reference's known-answer tests
 * (src/main.zig:1078-1150); orc_transformer end-to-end is "parity unpinned"
 * (no Zig compiler, no checkpoint, no golden vector in the reference).
 *
 * Build: make -C oracle   (gcc -O3 -march=x86-64-v3 -ffp-contract=off; v3 = AVX2+FMA so the .so built here also runs on the GPU box host)
 * -ffp-contract=off matters: fusing is decided by the use_fma knob, never by
 * the compiler.  Citations are to /root/reference/src/main.zig.
 */
#include "llama2_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

static int g_vw = 8, g_fma = 0, g_tree = 0;

void orc_set_mode(int vector_width, int use_fma, int tree_reduce)
{
    if (vector_width == 4 || vector_width == 8 || vector_width == 16) g_vw = vector_width;
    g_fma = use_fma ? 1 : 0;
    g_tree = tree_reduce ? 1 : 0;
}
void orc_get_mode(int *vector_width, int *use_fma, int *tree_reduce)
{
    if (vector_width) *vector_width = g_vw;
    if (use_fma) *use_fma = g_fma;
    if (tree_reduce) *tree_reduce = g_tree;
}

#define VW 4
#define SUF 4
#include "oracle_kernels.inc"
#undef VW
#undef SUF
#define VW 8
#define SUF 8
#include "oracle_kernels.inc"
#undef VW
#undef SUF
#define VW 16
#define SUF 16
#include "oracle_kernels.inc"
#undef VW
#undef SUF

#define DISPATCH(call4, call8, call16) \
    do {                               \
        if (g_vw == 4) { call4; }      \
        else if (g_vw == 8) { call8; } \
        else { call16; }               \
    } while (0)

void orc_rmsnorm(float *o, const float *x, const float *w, size_t n)
{
    DISPATCH(rmsnorm_4(o, x, w, n, g_fma, g_tree), rmsnorm_8(o, x, w, n, g_fma, g_tree),
             rmsnorm_16(o, x, w, n, g_fma, g_tree));
}

float orc_vector_dot_product(const float *x, const float *y, size_t n)
{
    float r;
    DISPATCH(r = vector_dot_product_4(x, y, n, g_fma, g_tree),
             r = vector_dot_product_8(x, y, n, g_fma, g_tree),
             r = vector_dot_product_16(x, y, n, g_fma, g_tree));
    return r;
}

void orc_matmul_fused(int N, float *const *outs, const float *x, const float *const *ws, size_t n,
                      size_t d)
{
    if (N == 1)
        DISPATCH(matmul_fused1_4(outs, x, ws, n, d, g_fma, g_tree),
                 matmul_fused1_8(outs, x, ws, n, d, g_fma, g_tree),
                 matmul_fused1_16(outs, x, ws, n, d, g_fma, g_tree));
    else if (N == 2)
        DISPATCH(matmul_fused2_4(outs, x, ws, n, d, g_fma, g_tree),
                 matmul_fused2_8(outs, x, ws, n, d, g_fma, g_tree),
                 matmul_fused2_16(outs, x, ws, n, d, g_fma, g_tree));
    else
        DISPATCH(matmul_fused3_4(outs, x, ws, n, d, g_fma, g_tree),
                 matmul_fused3_8(outs, x, ws, n, d, g_fma, g_tree),
                 matmul_fused3_16(outs, x, ws, n, d, g_fma, g_tree));
}

/* main.zig:485-498: matmul is matmul_fused(1, ...) */
void orc_matmul(float *xout, const float *x, const float *w, size_t n, size_t d)
{
    float *outs[1] = {xout};
    const float *ws[1] = {w};
    orc_matmul_fused(1, outs, x, ws, n, d);
}

/* main.zig:608-628: elementwise, lane order irrelevant */
void orc_vector_mul(float *x, const float *y, size_t n)
{
    for (size_t i = 0; i < n; i++) x[i] *= y[i];
}

/* main.zig:632-653 (only reached from a reference unit test) */
void orc_vector_weighted_sum(float *xout, const float *x, float y, size_t n)
{
    for (size_t i = 0; i < n; i++) xout[i] = g_fma ? fmaf(x[i], y, xout[i]) : xout[i] + x[i] * y;
}

void orc_vector_weighted_sum_rows(float *xout, size_t xout_len, const float *rows,
                                  size_t row_stride, const float *weights, size_t n_weights)
{
    vector_weighted_sum_rows_8(xout, xout_len, rows, row_stride, weights, n_weights, g_fma);
}

/* main.zig:687-706: scalar; divide by sum, not multiply by reciprocal */
void orc_softmax(float *x, size_t n)
{
    float max = x[0];
    for (size_t i = 1; i < n; i++)
        if (x[i] > max) max = x[i];
    float sum = 0.0f;
    for (size_t i = 0; i < n; i++) {
        x[i] = expf(x[i] - max);
        sum += x[i];
    }
    for (size_t i = 0; i < n; i++) x[i] /= sum;
}

/* main.zig:708-713 */
void orc_accum(float *a, const float *b, size_t n)
{
    for (size_t i = 0; i < n; i++) a[i] += b[i];
}

/* main.zig:715-726: strict '>' so the lowest index wins ties */
size_t orc_argmax(const float *x, size_t n)
{
    float max = x[0];
    size_t maxi = 0;
    for (size_t i = 1; i < n; i++)
        if (x[i] > max) {
            max = x[i];
            maxi = i;
        }
    return maxi;
}

/* ------------------------------------------------------------------ */

static size_t kv_dim_of(const orc_config *c) { return ((size_t)c->dim * c->n_kv_heads) / c->n_heads; }

/* main.zig:85-112: the pointer walk, in f32 units */
size_t orc_weights_count(const orc_config *c, int shared_weights)
{
    const size_t V = c->vocab_size, dim = c->dim, hid = c->hidden_dim, L = c->n_layers;
    const size_t H = c->n_heads, KV = c->n_kv_heads, S = c->seq_len, hs = dim / H;
    size_t n = 0;
    n += V * dim;                 /* token_embedding_table :86-87 */
    n += L * dim;                 /* rms_att_weight :88-89 */
    n += L * dim * (H * hs);      /* wq :90-91 */
    n += L * dim * (KV * hs);     /* wk :92-93 */
    n += L * dim * (KV * hs);     /* wv :94-95 */
    n += L * (H * hs) * dim;      /* wo :96-97 */
    n += L * dim;                 /* rms_ffn_weight :98-99 */
    n += L * dim * hid;           /* w1 :100-101 */
    n += L * hid * dim;           /* w2 :102-103 */
    n += L * dim * hid;           /* w3 :104-105 */
    n += dim;                     /* rms_final_weight :106-107 */
    n += S * hs / 2;              /* freq_cis_real :108-109 */
    n += S * hs / 2;              /* freq_cis_imag :110-111 */
    if (!shared_weights) n += V * dim; /* wcls :112 */
    return n;
}

void orc_weights_init(orc_weights *w, const orc_config *c, const float *data, int shared_weights)
{
    const size_t V = c->vocab_size, dim = c->dim, hid = c->hidden_dim, L = c->n_layers;
    const size_t H = c->n_heads, KV = c->n_kv_heads, S = c->seq_len, hs = dim / H;
    const float *ptr = data;
    w->token_embedding_table = ptr; ptr += V * dim;
    w->rms_att_weight = ptr;        ptr += L * dim;
    w->wq = ptr;                    ptr += L * dim * (H * hs);
    w->wk = ptr;                    ptr += L * dim * (KV * hs);
    w->wv = ptr;                    ptr += L * dim * (KV * hs);
    w->wo = ptr;                    ptr += L * (H * hs) * dim;
    w->rms_ffn_weight = ptr;        ptr += L * dim;
    w->w1 = ptr;                    ptr += L * dim * hid;
    w->w2 = ptr;                    ptr += L * hid * dim;
    w->w3 = ptr;                    ptr += L * dim * hid;
    w->rms_final_weight = ptr;      ptr += dim;
    w->freq_cis_real = ptr;         ptr += S * hs / 2;
    w->freq_cis_imag = ptr;         ptr += S * hs / 2;
    w->wcls = shared_weights ? w->token_embedding_table : ptr;
}

static float *falloc(size_t n)
{
    void *p = NULL;
    if (posix_memalign(&p, 64, (n ? n : 1) * sizeof(float)) != 0) return NULL;
    memset(p, 0, (n ? n : 1) * sizeof(float));
    return (float *)p;
}

/* main.zig:137-154 */
int orc_runstate_init(orc_runstate *s, const orc_config *c)
{
    const size_t kv_dim = kv_dim_of(c);
    memset(s, 0, sizeof *s);
    s->x = falloc(c->dim);
    s->xb = falloc(c->dim);
    s->xb2 = falloc(c->dim);
    s->hb = falloc(c->hidden_dim);
    s->hb2 = falloc(c->hidden_dim);
    s->q = falloc(c->dim);
    s->k = falloc(kv_dim);
    s->v = falloc(kv_dim);
    s->att = falloc((size_t)c->n_heads * c->seq_len);
    s->logits = falloc(c->vocab_size);
    s->key_cache = falloc((size_t)c->n_layers * c->seq_len * kv_dim);
    s->value_cache = falloc((size_t)c->n_layers * c->seq_len * kv_dim);
    if (!s->x || !s->xb || !s->xb2 || !s->hb || !s->hb2 || !s->q || !s->k || !s->v || !s->att ||
        !s->logits || !s->key_cache || !s->value_cache) {
        orc_runstate_free(s);
        return -1;
    }
    return 0;
}

void orc_runstate_free(orc_runstate *s)
{
    free(s->x); free(s->xb); free(s->xb2); free(s->hb); free(s->hb2); free(s->q);
    free(s->k); free(s->v); free(s->att); free(s->logits); free(s->key_cache);
    free(s->value_cache);
    memset(s, 0, sizeof *s);
}

/* RoPE, main.zig:336-351 (inline in transformer()): adjacent pairs (i, i+1), freq from pow, cos,
 * sin in f32, recomputed per pair; k is rotated only where i < kv_dim (:343). */
void orc_rope(float *q, float *k, size_t pos, size_t dim, size_t kv_dim, size_t head_size)
{
    for (size_t i = 0; i < dim; i += 2) {
        const float head_dim = (float)(i % head_size);                        /* :338 */
        const float freq = 1.0f / powf(10000.0f, head_dim / (float)head_size); /* :339 */
        const float val = (float)pos * freq;                                  /* :340 */
        const float fcr = cosf(val), fci = sinf(val);                         /* :341-342 */
        const int rotn = i < kv_dim ? 2 : 1;                                  /* :343 */
        for (int v = 0; v < rotn; v++) {
            float *vec = v == 0 ? q : k;
            const float v0 = vec[i], v1 = vec[i + 1];
            vec[i] = v0 * fcr - v1 * fci;                                     /* :348 */
            vec[i + 1] = v0 * fci + v1 * fcr;                                 /* :349 */
        }
    }
}

/* main.zig:285-430 */
void orc_transformer(size_t token, size_t pos, const orc_config *c, orc_runstate *s,
                     const orc_weights *w)
{
    const size_t dim = c->dim, hidden_dim = c->hidden_dim;
    const size_t head_size = dim / c->n_heads;                  /* :289 */
    const size_t kv_dim = kv_dim_of(c);                         /* :290 */
    const size_t kv_mul = c->n_heads / c->n_kv_heads;           /* :291 */
    const size_t seq_len = c->seq_len;
    float *x = s->x;

    memcpy(x, w->token_embedding_table + token * dim, dim * sizeof(float)); /* :295-296 */

    for (size_t l = 0; l < (size_t)c->n_layers; l++) {          /* :303 */
        orc_rmsnorm(s->xb, x, w->rms_att_weight + l * dim, dim); /* :305 */

        if (kv_dim == dim) {                                    /* :308-313 */
            float *outs[3] = {s->q, s->k, s->v};
            const float *ws[3] = {w->wq + l * dim * dim, w->wk + l * dim * kv_dim,
                                  w->wv + l * dim * kv_dim};
            orc_matmul_fused(3, outs, s->xb, ws, dim, dim);
        } else {                                                /* :315-319 */
            orc_matmul(s->q, s->xb, w->wq + l * dim * dim, dim, dim);
            float *outs[2] = {s->k, s->v};
            const float *ws[2] = {w->wk + l * dim * kv_dim, w->wv + l * dim * kv_dim};
            orc_matmul_fused(2, outs, s->xb, ws, dim, kv_dim);
        }

        orc_rope(s->q, s->k, pos, dim, kv_dim, head_size);      /* :336-351 */

        const size_t loff = l * seq_len * kv_dim;               /* :354 */
        memcpy(s->key_cache + loff + pos * kv_dim, s->k, kv_dim * sizeof(float));   /* :357 */
        memcpy(s->value_cache + loff + pos * kv_dim, s->v, kv_dim * sizeof(float)); /* :358 */

        for (size_t h = 0; h < (size_t)c->n_heads; h++) {       /* :361 */
            const float *q = s->q + h * head_size;
            float *att = s->att + h * seq_len;
            for (size_t t = 0; t <= pos; t++) {                 /* :367-375 */
                const float *k = s->key_cache + loff + t * kv_dim + (h / kv_mul) * head_size;
                float score = orc_vector_dot_product(q, k, head_size);
                score /= sqrtf((float)head_size);               /* :372: divide */
                att[t] = score;
            }
            orc_softmax(att, pos + 1);                          /* :378 */
            orc_vector_weighted_sum_rows(s->xb + h * head_size, head_size,
                                         s->value_cache + loff + (h / kv_mul) * head_size, kv_dim,
                                         att, pos + 1);         /* :381-388 */
        }

        orc_matmul(s->xb2, s->xb, w->wo + l * dim * dim, dim, dim); /* :392 */
        orc_accum(x, s->xb2, dim);                                  /* :395 */
        orc_rmsnorm(s->xb, x, w->rms_ffn_weight + l * dim, dim);    /* :398 */

        {                                                           /* :405-408 */
            float *outs[2] = {s->hb, s->hb2};
            const float *ws[2] = {w->w1 + l * dim * hidden_dim, w->w3 + l * dim * hidden_dim};
            orc_matmul_fused(2, outs, s->xb, ws, dim, hidden_dim);
        }
        for (size_t i = 0; i < hidden_dim; i++)                     /* :411-413 */
            s->hb[i] = s->hb[i] * (1.0f / (1.0f + expf(-s->hb[i])));
        orc_vector_mul(s->hb, s->hb2, hidden_dim);                  /* :416 */
        orc_matmul(s->xb, s->hb, w->w2 + l * dim * hidden_dim, hidden_dim, dim); /* :419 */
        orc_accum(x, s->xb, dim);                                   /* :422 */
    }

    orc_rmsnorm(x, x, w->rms_final_weight, dim);                    /* :426 */
    orc_matmul(s->logits, x, w->wcls, dim, c->vocab_size);          /* :429 */
}

/* main.zig:987-1042 with temperature == 0.0 (:1002-1003) */
size_t orc_generate_greedy(const orc_config *c, orc_runstate *s, const orc_weights *w,
                           const int32_t *prompt, size_t n_prompt, size_t steps,
                           int32_t *out_tokens, float *margins)
{
    size_t token = 1;                                   /* :988 BOS */
    size_t seq_len = steps == 0 ? (size_t)c->seq_len : steps;   /* :992 */
    if (seq_len < 1) seq_len = 1;
    if (seq_len > (size_t)c->seq_len) seq_len = c->seq_len;     /* :993 */
    size_t pos = 0, n = 0;
    for (; pos < seq_len; pos++) {                      /* :995 */
        orc_transformer(token, pos, c, s, w);           /* :996 */
        size_t next;
        if (pos < n_prompt) next = (size_t)prompt[pos]; /* :999-1000 */
        else next = orc_argmax(s->logits, c->vocab_size);
        if (margins) {
            float m1 = -INFINITY, m2 = -INFINITY;
            for (size_t i = 0; i < (size_t)c->vocab_size; i++) {
                const float v = s->logits[i];
                if (v > m1) { m2 = m1; m1 = v; }
                else if (v > m2) m2 = v;
            }
            margins[n] = m1 - m2;
        }
        out_tokens[n++] = (int32_t)next;
        if (next == 1) break;                           /* :1017 */
        token = next;                                   /* :1036 */
    }
    return n;
}

/* ------------------------------------------------------------------ */
/* Seeded synthetic checkpoints.  Not part of the reference. */

static inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

float orc_synth_value(uint64_t idx, uint64_t seed, float scale, float bias)
{
    const uint64_t z = mix64(idx + seed * 0x9E3779B97F4A7C15ULL);
    const uint32_t u = (uint32_t)(z >> 41);             /* 23 bits */
    const float r = (float)u * 0x1p-22f - 1.0f;         /* exact, [-1,1) */
    return bias + scale * r;                            /* contract=off: mul then add */
}

void orc_synth_fill_range(float *dst, uint64_t base_idx, uint64_t count, uint64_t seed,
                          float scale, float bias)
{
    for (uint64_t i = 0; i < count; i++) dst[i] = orc_synth_value(base_idx + i, seed, scale, bias);
}

typedef struct {
    float *dst;
    uint64_t base, count, seed;
    float scale, bias;
} fill_job;

static void *fill_thread(void *p)
{
    fill_job *j = (fill_job *)p;
    orc_synth_fill_range(j->dst, j->base, j->count, j->seed, j->scale, j->bias);
    return NULL;
}

static void fill_parallel(float *dst, uint64_t base, uint64_t count, uint64_t seed, float scale,
                          float bias, int n_threads)
{
    if (n_threads <= 1 || count < (1u << 20)) {
        orc_synth_fill_range(dst, base, count, seed, scale, bias);
        return;
    }
    if (n_threads > 64) n_threads = 64;
    pthread_t th[64];
    fill_job jobs[64];
    const uint64_t chunk = (count + n_threads - 1) / n_threads;
    int started = 0;
    for (int t = 0; t < n_threads; t++) {
        const uint64_t lo = (uint64_t)t * chunk;
        if (lo >= count) break;
        const uint64_t hi = lo + chunk > count ? count : lo + chunk;
        jobs[t] = (fill_job){dst + lo, base + lo, hi - lo, seed, scale, bias};
        pthread_create(&th[t], NULL, fill_thread, &jobs[t]);
        started++;
    }
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
}

/* Per-tensor (scale,bias); keep in sync with csrc/misc_kernels.hip: synth_fill_kernel and checkpoint.py:
 *   matrices (d,n)         : uniform +-sqrt(3/n)  -> unit-variance outputs
 *   token_embedding / wcls : uniform +-2*sqrt(3/dim)
 *   rmsnorm weights        : 1 + 0.1*r
 *   freq_cis gap           : r (never read)
 */
void orc_synth_fill(float *data, const orc_config *c, int shared_weights, uint64_t seed,
                    int n_threads)
{
    const uint64_t V = c->vocab_size, dim = c->dim, hid = c->hidden_dim, L = c->n_layers;
    const uint64_t S = c->seq_len, hs = dim / c->n_heads, kvd = kv_dim_of(c);
    const float s_dim = sqrtf(3.0f / (float)dim), s_hid = sqrtf(3.0f / (float)hid);
    const float s_emb = 2.0f * s_dim;
    uint64_t o = 0;
#define T(count, scale, bias)                                                    \
    do {                                                                         \
        fill_parallel(data + o, o, (count), seed, (scale), (bias), n_threads);   \
        o += (count);                                                            \
    } while (0)
    T(V * dim, s_emb, 0.0f);        /* token_embedding_table */
    T(L * dim, 0.1f, 1.0f);         /* rms_att_weight */
    T(L * dim * dim, s_dim, 0.0f);  /* wq */
    T(L * kvd * dim, s_dim, 0.0f);  /* wk */
    T(L * kvd * dim, s_dim, 0.0f);  /* wv */
    T(L * dim * dim, s_dim, 0.0f);  /* wo */
    T(L * dim, 0.1f, 1.0f);         /* rms_ffn_weight */
    T(L * hid * dim, s_dim, 0.0f);  /* w1 (hidden,dim): n = dim */
    T(L * dim * hid, s_hid, 0.0f);  /* w2 (dim,hidden): n = hidden */
    T(L * hid * dim, s_dim, 0.0f);  /* w3 */
    T(dim, 0.1f, 1.0f);             /* rms_final_weight */
    T(S * hs / 2, 1.0f, 0.0f);      /* freq_cis_real (unused) */
    T(S * hs / 2, 1.0f, 0.0f);      /* freq_cis_imag (unused) */
    if (!shared_weights) T(V * dim, s_emb, 0.0f); /* wcls */
#undef T
}
