// hooks.cpp -- kernel-level entry points of include/llama2_hip_test.h (tests, and the reference's own
// unit-test shapes).  What runs behind each hook:
//   l2z_matmul / l2z_matmul_fused   launch_matvec: the forward pass's own kernels for that width
//   l2z_rmsnorm                     the mat-vec prologue's staging code (xload_issue / xstage_finish)
//   l2z_attention_decode            the forward pass's attention kernels, every form selectable
//   l2z_prefill_attention           the batched prefill's attention kernels, every form selectable
//   l2z_softmax, l2z_vector_dot_product, l2z_vector_weighted_sum_rows
//                                   the GENERIC attention kernel's device functions (block_softmax,
//                                   attn_scores, attn_weighted_sum) -- the forms the stories / 7B
//                                   shapes run are covered by l2z_attention_decode instead
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "l2z_state.h"

using namespace l2z;

// ---------------------------------------------------------------------------
// Kernel-level test hooks: upload, launch, download.
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

// The reference's cache order (seq_len, kv_dim) (main.zig:354) -> the device's head-major order
// [kv head][seq_len][head_size], so that the hooks drive the kernels on the layout the forward
// pass gives them.
static std::vector<float> to_head_major(const float *src, int seq_len, int n_kv_heads, int head_size)
{
    const size_t kvd = (size_t)n_kv_heads * head_size;
    std::vector<float> out((size_t)seq_len * kvd);
    for (int h = 0; h < n_kv_heads; h++)
        for (int t = 0; t < seq_len; t++)
            memcpy(out.data() + ((size_t)h * seq_len + t) * head_size, src + (size_t)t * kvd + (size_t)h * head_size,
                   (size_t)head_size * sizeof(float));
    return out;
}

// src/main.zig:361-389 for the P queries of a prompt chunk at positions pos0 .. pos0 + P - 1, through the
// batched prefill's attention kernels (prefill_attention.hip).  form: 0 as l2z_prefill picks, 1 block per
// (head, query), 2 tiled with the softmax in LDS, 3 flash form (head sizes 64 and 128).
extern "C" int l2z_prefill_attention(int form, float *out, const float *q, const float *kcache, const float *vcache,
                                     int pos0, int n_queries, int n_heads, int n_kv_heads, int head_size, int seq_len)
{
    L2Z_CHECK(out && q && kcache && vcache && n_heads > 0 && n_kv_heads > 0 && head_size > 0 && head_size % 4 == 0 &&
                  n_heads % n_kv_heads == 0 && n_queries > 0 && pos0 >= 0 && pos0 + n_queries <= seq_len,
              L2Z_ERR_INVALID, "l2z_prefill_attention: bad arguments");
    L2Z_CHECK(form >= 0 && form <= 3, L2Z_ERR_INVALID, "l2z_prefill_attention: form %d", form);
    L2Z_CHECK(form < 3 || head_size == 64 || head_size == 128, L2Z_ERR_INVALID,
              "l2z_prefill_attention: the flash form takes head sizes 64 and 128");
    L2Z_TRY(ensure_device(current_device_for(nullptr)));
    const size_t dim = (size_t)n_heads * head_size, kvd = (size_t)n_kv_heads * head_size;
    DevBuf dq, dk, dv, dout;
    L2Z_TRY(dq.alloc(n_queries * dim)); L2Z_TRY(dk.alloc(seq_len * kvd)); L2Z_TRY(dv.alloc(seq_len * kvd));
    L2Z_TRY(dout.alloc(n_queries * dim));
    const std::vector<float> hk = to_head_major(kcache, seq_len, n_kv_heads, head_size),
                             hv = to_head_major(vcache, seq_len, n_kv_heads, head_size);
    L2Z_TRY(dq.up(q, n_queries * dim)); L2Z_TRY(dk.up(hk.data(), seq_len * kvd)); L2Z_TRY(dv.up(hv.data(), seq_len * kvd));
    const hipError_t e = launch_prefill_attention(dq.p, (int)dim, dk.p, dv.p, dout.p, (int)dim, pos0, n_queries, n_heads,
                                                  head_size, head_size, (size_t)seq_len * head_size, n_heads / n_kv_heads,
                                                  seq_len, nullptr, n_heads, form);
    L2Z_HIP(e);
    L2Z_HIP(hipDeviceSynchronize());
    L2Z_TRY(dout.down(out, n_queries * dim));
    return L2Z_OK;
}

// src/main.zig:361-389 for one layer, through the forward pass's attention kernels
extern "C" int l2z_attention_decode(int form, int nch, float *out, const float *q, const float *kcache,
                                    const float *vcache, int pos, int n_heads, int n_kv_heads,
                                    int head_size, int seq_len)
{
    L2Z_CHECK(out && q && kcache && vcache && n_heads > 0 && n_kv_heads > 0 && head_size > 0 &&
                  n_heads % n_kv_heads == 0 && seq_len > 0 && pos >= 0 && pos < seq_len,
              L2Z_ERR_INVALID, "l2z_attention_decode: bad arguments");
    L2Z_CHECK(form >= 0 && form <= 4, L2Z_ERR_INVALID, "l2z_attention_decode: form %d", form);
    L2Z_TRY(ensure_device(current_device_for(nullptr)));
    const size_t dim = (size_t)n_heads * head_size, kvd = (size_t)n_kv_heads * head_size;
    const size_t kvn = (size_t)seq_len * kvd;
    DevBuf dq, dk, dv, dout, dpart;
    int *dpos = nullptr, *dcnt = nullptr;
    L2Z_TRY(dq.alloc(dim)); L2Z_TRY(dk.alloc(kvn)); L2Z_TRY(dv.alloc(kvn)); L2Z_TRY(dout.alloc(dim));
    const std::vector<float> hk = to_head_major(kcache, seq_len, n_kv_heads, head_size),
                             hv = to_head_major(vcache, seq_len, n_kv_heads, head_size);
    L2Z_TRY(dq.up(q, dim)); L2Z_TRY(dk.up(hk.data(), kvn)); L2Z_TRY(dv.up(hv.data(), kvn));
    AttnArgs a = {};
    a.q = dq.p; a.kcache = dk.p; a.vcache = dv.p; a.xb = dout.p;
    a.head_size = head_size; a.kv_row = head_size; a.kv_head = (size_t)seq_len * head_size;
    a.kv_mul = n_heads / n_kv_heads; a.seq_len = seq_len;
    const bool fast_ok = attention_split_supported(a);  // same conditions as the fast kernels
    L2Z_CHECK(form == 0 || form == 4 || fast_ok, L2Z_ERR_INVALID,
              "l2z_attention_decode: form %d needs head_size %% 4 == 0 and <= 256", form);
    const size_t att_lds = attention_lds_bytes(head_size, seq_len, fast_ok);
    L2Z_CHECK(att_lds <= 160 * 1024, L2Z_ERR_INVALID, "l2z_attention_decode: seq_len too long for LDS");
    hipError_t e = hipMalloc(&dpos, sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(dpos, &pos, sizeof(int), hipMemcpyHostToDevice);
    a.pos_ptr = dpos;
    int use_nch = 0;
    if (e == hipSuccess && (form == 3 || (form == 0 && fast_ok && pos >= 256 && seq_len > 256))) {
        use_nch = nch > 0 ? nch : attention_split_chunks(n_heads, g_cus);
        if (use_nch > 16) use_nch = 16;
        if (use_nch < 2) use_nch = form == 3 ? 2 : 0;
    }
    if (e == hipSuccess && use_nch > 1) {
        int rc = dpart.alloc(attention_split_part_floats(n_heads, head_size, use_nch));
        if (rc != L2Z_OK) { (void)hipFree(dpos); return rc; }
        e = hipMalloc(&dcnt, (size_t)n_heads * sizeof(int));
        if (e == hipSuccess) e = hipMemset(dcnt, 0, (size_t)n_heads * sizeof(int));
        // twice: the second launch finds the counters as the first one left them
        const bool small = pos < attention_split_wide_pos(seq_len);  // as the forward pass picks at this position
        if (e == hipSuccess) e = launch_attention_split(a, n_heads, use_nch, dpart.p, dcnt, nullptr, small);
        if (e == hipSuccess) e = launch_attention_split(a, n_heads, use_nch, dpart.p, dcnt, nullptr, small);
    } else if (e == hipSuccess) {
        // form 0: what the forward pass launches at this position (attn_variant: the short-context form first)
        const int auto_form = pos < attention_short_pos(head_size, seq_len) ? 1 : 0;
        e = launch_attention(a, n_heads, nullptr, form == 3 || form == 0 ? auto_form : form);
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)hipFree(dpos);
    if (dcnt) (void)hipFree(dcnt);
    L2Z_HIP(e);
    return dout.down(out, dim);
}

extern "C" int l2z_option_set(const char *env_name, long long value)
{
    L2Z_CHECK(env_name != nullptr && tunables_set(env_name, value), L2Z_ERR_INVALID,
              "l2z_option_set: unknown option '%s'", env_name ? env_name : "(null)");
    return L2Z_OK;
}

// Host-side shard geometry (no device needed): what rank `rank` of `world` owns under scheme A (rows) and, for Wo / W2 under
// scheme B, the padded width of its column shards.  out[0..9] = dim0, dim_loc, kvd_loc, heads_loc, hid0, hid_loc, v0, v_loc,
// dimc_pad, hidc_pad.
extern "C" int l2z_shard_plan(const l2z_config *config, int rank, int world, int *out, int cap)
{
    L2Z_CHECK(config != nullptr && out != nullptr && cap >= 10 && world >= 1 && rank >= 0 && rank < world, L2Z_ERR_INVALID,
              "l2z_shard_plan: bad arguments");
    l2z_comm c;
    c.rank = rank; c.world = world; c.device = 0; c.nccl = nullptr;
    Shard sh;
    L2Z_TRY(make_shard(*config, &c, &sh));
    const int v[10] = {sh.dim0, sh.dim_loc, sh.kvd_loc, sh.heads_loc, sh.hid0, sh.hid_loc, sh.v0, sh.v_loc, sh.dimc_pad, sh.hidc_pad};
    for (int i = 0; i < 10; i++) out[i] = v[i];
    return L2Z_OK;
}

// Host-side planning of the batched prefill, no device needed: how a prompt is cut into chunks, and which
// output tile the direct-to-LDS GEMM takes for a [P, N] product (0: 128x64, 1: 64x64, 2: 32x64, 3: 32x32,
// 4: 128x128; the CU count is the current device's, 256 without one).
extern "C" int l2z_prefill_plan_model(const l2z_config *config, int n_tokens, int *chunks, int cap)
{
    L2Z_CHECK(config != nullptr && n_tokens >= 0 && (chunks != nullptr || cap == 0), L2Z_ERR_INVALID, "l2z_prefill_plan_model: bad arguments");
    L2Z_CHECK(config->n_heads > 0 && config->dim > 0, L2Z_ERR_INVALID, "l2z_prefill_plan_model: bad config");
    int n = 0;
    for (int done = 0; done < n_tokens; n++) {
        const int P = prefill_next_chunk_of(*config, n_tokens - done);
        if (n < cap) chunks[n] = P;
        done += P;
    }
    return n;
}

extern "C" int l2z_prefill_plan(int n_tokens, int *chunks, int cap)
{
    L2Z_CHECK(n_tokens >= 0 && (chunks != nullptr || cap == 0), L2Z_ERR_INVALID, "l2z_prefill_plan: bad arguments");
    int n = 0;
    for (int done = 0; done < n_tokens; n++) {
        const int P = prefill_next_chunk(n_tokens - done);
        if (n < cap) chunks[n] = P;
        done += P;
    }
    return n;
}

extern "C" int l2z_prefill_split_k(long long n_features_whole, int n_tokens, int k, int paired)
{
    L2Z_CHECK(n_features_whole > 0 && n_tokens > 0 && k > 0, L2Z_ERR_INVALID, "l2z_prefill_split_k: bad arguments");
    return prefill_split_k(n_features_whole, n_tokens, k, paired != 0);
}

extern "C" int l2z_prefill_cores(long long n_features_whole, int n_tokens, int k)
{
    L2Z_CHECK(n_features_whole > 0 && n_tokens > 0 && k > 0, L2Z_ERR_INVALID, "l2z_prefill_cores: bad arguments");
    const int kp = (k + 63) / 64 * 64;
    if (!x3_applies(n_features_whole, kp)) return 0;
    return x3_stream_shape(n_features_whole, n_tokens, kp) ? 1 + x3_stream_sk(n_features_whole, n_tokens, kp) : 1;
}

extern "C" int l2z_prefill_tile(int n_features, int n_tokens, int paired)
{
    L2Z_CHECK(n_features > 0 && n_tokens > 0, L2Z_ERR_INVALID, "l2z_prefill_tile: bad arguments");
    return prefill_tile_form(n_features, n_tokens, paired);
}
