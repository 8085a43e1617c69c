// weights.cpp -- shard geometry, the llama2.c-v0 tensor table (main.zig:85-112) and the device
// resident Weights object of the C ABI (include/llama2_hip.h).
//
// Product code.  No CPU fallback anywhere: without a HIP device every compute entry point returns
// L2Z_ERR_NO_DEVICE.  Nothing under oracle/ is referenced.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "l2z_state.h"
#include "tunables.h"

namespace l2z {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

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
    // (a 1-rank RCCL communicator takes it too: the ncclAllReduce call path, testable on one GPU)
    s.scheme_b = tunables().scheme_b != 0 && comm != nullptr && (s.world > 1 || comm->nccl != nullptr);
    s.dimc_pad = pad_cols(s.dim_loc);
    s.hidc_pad = pad_cols(s.hid_loc);
    *out = s;
    return L2Z_OK;
}

int pad_cols(int n)
{
    const int n4 = (n + 3) / 4;
    return n > 768 ? ((n4 + 63) / 64) * 256 : n4 * 4;
}

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

}  // namespace l2z

using namespace l2z;

namespace {

constexpr size_t kBlobSlackFloats = 1024;
bool is_w1(const TensorDesc &d) { return strcmp(d.name, "w1") == 0; }
bool is_w3(const TensorDesc &d) { return strcmp(d.name, "w3") == 0; }

// scheme B: Wo / W2 live here as this rank's COLUMNS of every row, each row padded with zeros to cpad floats
bool col_shard(const l2z_weights *w, const TensorDesc &d, size_t *c0, size_t *cl, size_t *cpad)
{
    if (!w->sh.scheme_b || d.kind != BY_DIM_ROWS) return false;
    const bool is_wo = strcmp(d.name, "wo") == 0;
    *c0 = is_wo ? w->sh.dim0 : w->sh.hid0;
    *cl = is_wo ? w->sh.dim_loc : w->sh.hid_loc;
    *cpad = is_wo ? w->sh.dimc_pad : w->sh.hidc_pad;
    return true;
}

// floats a tensor takes in the device blob: this rank's rows; W1's slot also holds W3's rows (interleaved),
// W3 has none of its own; the never-read freq_cis tables stay resident only in the unsharded layout (weights_read)
size_t slot_floats(const l2z_weights *w, const TensorDesc &d)
{
    size_t r0, r1;
    shard_rows(d, w->sh, &r0, &r1);
    if (d.kind == SKIP && !w->file_layout) return 0;
    if (is_w3(d)) return 0;
    size_t c0, cl, cpad;
    if (col_shard(w, d, &c0, &cl, &cpad)) return d.layers * d.rows * cpad;
    return d.layers * (r1 - r0) * d.cols * (is_w1(d) ? 2 : 1);
}

void carve_local(l2z_weights *w, const std::vector<TensorDesc> &tt)
{
    // device layout: the file's tensor order, each tensor (layers, rows_loc, cols); W1 | W3 row-interleaved in W1's slot
    size_t off = 0, off_w1 = 0;
    const float *base = w->blob;
    w->dev_off.assign(tt.size(), 0);
    for (size_t i = 0; i < tt.size(); i++) {
        const auto &d = tt[i];
        if (is_w1(d)) off_w1 = off;
        w->dev_off[i] = is_w3(d) ? off_w1 + d.cols : off;
        const float *p = base + w->dev_off[i];
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
        off += slot_floats(w, d);
    }
    if (w->shared) w->wcls = w->tok_emb + (size_t)w->sh.v0 * w->cfg.dim;  // main.zig:112
}

size_t local_floats(const l2z_weights *w, const std::vector<TensorDesc> &tt)
{
    size_t off = 0;
    for (const auto &d : tt) off += slot_floats(w, d);
    return off;
}

// Host -> device copy of weight rows (main.zig:936-967 reads the whole file into the heap first; here
// the source is the caller's buffer, typically the mmapped checkpoint).  A plain hipMemcpy from the
// pageable mapping: the runtime stages it through its own pinned buffers at 54-56 GB/s for the
// 27 GB llama2-7b file (profiles/r02_upload_rate.txt).  An explicit pinned double buffer filled by
// eight host threads was measured beside it and was no faster (46-53 GB/s), so it is not kept.
hipError_t upload(float *dst, const float *src, size_t n_floats)
{
    const size_t piece = (size_t)256 << 20;  // floats: 1 GiB per call
    hipError_t e = hipSuccess;
    for (size_t o = 0; o < n_floats && e == hipSuccess; o += piece) {
        const size_t n = n_floats - o < piece ? n_floats - o : piece;
        e = hipMemcpy(dst + o, src + o, n * sizeof(float), hipMemcpyHostToDevice);
    }
    return e;
}

int weights_alloc(const l2z_config *config, int shared_weights, const l2z_comm *comm,
                  l2z_weights **out, std::vector<TensorDesc> *tt_out)
{
    L2Z_CHECK(config != nullptr && out != nullptr, L2Z_ERR_INVALID, "weights_init: null argument");
    const int dev = current_device_for(comm);
    L2Z_TRY(ensure_device(dev));
    Shard sh;
    L2Z_TRY(make_shard(*config, comm, &sh));
    static std::atomic<uint64_t> next_uid{1};
    l2z_weights *w = new l2z_weights();
    w->uid = next_uid.fetch_add(1);
    w->cfg = *config;
    w->shared = shared_weights ? 1 : 0;
    w->device = dev;
    w->sh = sh;
    w->file_layout = sh.world == 1 && !sh.scheme_b;
    *tt_out = tensor_table(*config, w->shared != 0);
    w->blob_floats = local_floats(w, *tt_out);
    // + a zeroed slack: the batched prefill's GEMMs multiply whole stages of K and read up to 3 x 256 floats past the
    // end of a W row against zero activation columns (prefill_common.h pad_k); past the LAST row of the LAST tensor
    // that is here (everywhere else it is the next row or the next tensor: finite)
    hipError_t e = hipMalloc(&w->blob, (w->blob_floats + kBlobSlackFloats) * sizeof(float));
    if (e == hipSuccess) e = hipMemset(w->blob + w->blob_floats, 0, kBlobSlackFloats * sizeof(float));
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
    // this rank's rows of every tensor, straight from the caller's buffer (typically the mmapped checkpoint);
    // W1 / W3 rows go through a device staging buffer of one layer and are spread over the shared slot there
    hipError_t e = hipSuccess;
    float *stage = nullptr;
    for (size_t i = 0; i < tt.size() && e == hipSuccess; i++) {
        const auto &d = tt[i];
        if (d.kind == SKIP && !w->file_layout) continue;
        size_t r0, r1;
        shard_rows(d, w->sh, &r0, &r1);
        const size_t rl = r1 - r0;
        float *dst = w->blob + w->dev_off[i];
        size_t c0, cl, cpad;
        if (col_shard(w, d, &c0, &cl, &cpad)) {
            // columns [c0, c0 + cl) of every row; the pad columns stay zero
            e = hipMemset(dst, 0, d.layers * d.rows * cpad * sizeof(float));
            for (size_t l = 0; l < d.layers && e == hipSuccess; l++)
                e = hipMemcpy2D(dst + l * d.rows * cpad, cpad * sizeof(float), data + d.offset + l * d.rows * d.cols + c0,
                                d.cols * sizeof(float), cl * sizeof(float), d.rows, hipMemcpyHostToDevice);
            continue;
        }
        if (is_w1(d) || is_w3(d)) {
            // in pieces of <= 64 MB through TWO staging buffers: a whole layer (180 MB at the 7B shape, ~1 GB at wider
            // ones) allocated after the blob could be the allocation that no longer fits, and a device-wide
            // synchronise per layer serialised the load.  A staging buffer is written again two pieces later; the
            // host-to-device copy and the spreading kernel both run in the null stream, in order.
            const size_t piece_rows = std::max<size_t>(1, std::min<size_t>(rl, ((size_t)64 << 20) / (d.cols * sizeof(float))));
            if (stage == nullptr) e = hipMalloc((void **)&stage, 2 * piece_rows * d.cols * sizeof(float));
            size_t n_piece = 0;
            for (size_t l = 0; l < d.layers && e == hipSuccess; l++) {
                for (size_t r = 0; r < rl && e == hipSuccess; r += piece_rows, n_piece++) {
                    const size_t nr = std::min(piece_rows, rl - r);
                    float *sb = stage + (n_piece & 1) * piece_rows * d.cols;
                    e = upload(sb, data + d.offset + (l * d.rows + r0 + r) * d.cols, nr * d.cols);
                    if (e == hipSuccess) e = launch_copy_rows(dst + (l * rl + r) * 2 * d.cols, 2 * d.cols, sb, nr, d.cols, nullptr);
                }
            }
            if (e == hipSuccess) e = hipDeviceSynchronize();
        } else if (rl == d.rows) {
            e = upload(dst, data + d.offset, d.count());
        } else {
            for (size_t l = 0; l < d.layers && e == hipSuccess; l++)
                e = upload(dst + l * rl * d.cols, data + d.offset + (l * d.rows + r0) * d.cols, rl * d.cols);
        }
    }
    if (stage) (void)hipFree(stage);
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
    for (size_t i = 0; i < tt.size(); i++) {
        const auto &d = tt[i];
        if (d.kind == SKIP && !w->file_layout) continue;
        size_t r0, r1;
        shard_rows(d, w->sh, &r0, &r1);
        const size_t rl = r1 - r0;
        const bool pair = is_w1(d) || is_w3(d);  // rows 2 * cols apart in the shared slot
        size_t c0, cl, cpad;
        if (col_shard(w, d, &c0, &cl, &cpad)) {
            float *dst = w->blob + w->dev_off[i];
            e = hipMemsetAsync(dst, 0, d.layers * d.rows * cpad * sizeof(float), nullptr);
            for (size_t l = 0; l < d.layers && e == hipSuccess; l++)
                e = launch_synth_fill(dst + l * d.rows * cpad, d.offset + l * d.rows * d.cols + c0, d.rows * cl, seed, d.scale, d.bias,
                                      nullptr, cl, cpad, d.cols);
            continue;
        }
        for (size_t l = 0; l < d.layers && e == hipSuccess; l++) {
            const uint64_t base = d.offset + (l * d.rows + r0) * d.cols;
            e = launch_synth_fill(w->blob + w->dev_off[i] + l * rl * d.cols * (pair ? 2 : 1), base, rl * d.cols, seed, d.scale,
                                  d.bias, nullptr, pair ? d.cols : 0, pair ? 2 * d.cols : 0);
        }
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

// offset / count address the FILE's order (main.zig:85-112); the device blob differs from it in W1 / W3 (rows
// interleaved in one slot), so the range is served tensor by tensor and, there, row by row
extern "C" int l2z_weights_read(const l2z_weights *w, size_t offset, size_t count, float *out)
{
    L2Z_CHECK(w != nullptr && out != nullptr, L2Z_ERR_INVALID, "l2z_weights_read: null argument");
    L2Z_CHECK(w->file_layout, L2Z_ERR_INVALID, "l2z_weights_read: only for unsharded weights");
    L2Z_CHECK(offset + count <= w->blob_floats, L2Z_ERR_INVALID, "l2z_weights_read: out of range");
    L2Z_HIP(hipSetDevice(w->device));
    const std::vector<TensorDesc> tt = tensor_table(w->cfg, w->shared != 0);
    const size_t lo = offset, hi = offset + count;
    for (size_t i = 0; i < tt.size(); i++) {
        const auto &d = tt[i];
        const size_t t0 = std::max(lo, d.offset), t1 = std::min(hi, d.offset + d.count());
        if (t0 >= t1) continue;
        if (!(is_w1(d) || is_w3(d))) {
            L2Z_HIP(hipMemcpy(out + (t0 - lo), w->blob + w->dev_off[i] + (t0 - d.offset), (t1 - t0) * sizeof(float),
                              hipMemcpyDeviceToHost));
            continue;
        }
        for (size_t f = t0; f < t1;) {  // row r of the tensor sits 2 * cols * r into the slot
            const size_t r = (f - d.offset) / d.cols, c = (f - d.offset) % d.cols;
            const float *src = w->blob + w->dev_off[i] + r * 2 * d.cols + c;
            const size_t whole = c == 0 ? (t1 - f) / d.cols : 0;  // whole rows from here: one strided copy
            if (whole > 0) {
                L2Z_HIP(hipMemcpy2D(out + (f - lo), d.cols * sizeof(float), src, 2 * d.cols * sizeof(float),
                                    d.cols * sizeof(float), whole, hipMemcpyDeviceToHost));
                f += whole * d.cols;
            } else {  // a partial row at either end of the range
                const size_t n = std::min(d.cols - c, t1 - f);
                L2Z_HIP(hipMemcpy(out + (f - lo), src, n * sizeof(float), hipMemcpyDeviceToHost));
                f += n;
            }
        }
    }
    return L2Z_OK;
}

extern "C" void l2z_weights_free(l2z_weights *w)
{
    if (!w) return;
    if (w->blob) (void)hipFree(w->blob);
    delete w;
}
