// llama2_host.cpp -- see llama2_host.hpp.  Citations: /root/reference/src/main.zig.
#include "llama2_host.hpp"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace l2zhost {

bool Tokenizer::from_file(const std::string &path, size_t vocab_size, std::string *err)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) {
        if (err) *err = "cannot open tokenizer '" + path + "'";
        return false;
    }
    auto fail = [&](const char *what) {
        if (err) *err = std::string("tokenizer '") + path + "': " + what;
        fclose(f);
        return false;
    };
    tokens.assign(vocab_size, std::string());
    scores.assign(vocab_size, 0.0f);
    first_index_.clear();
    if (fread(&max_token_len, sizeof(uint32_t), 1, f) != 1) return fail("truncated header");  // :186
    for (size_t i = 0; i < vocab_size; i++) {                                                 // :188
        uint32_t len = 0;
        if (fread(&scores[i], sizeof(float), 1, f) != 1) return fail("truncated score");      // :189
        if (fread(&len, sizeof(uint32_t), 1, f) != 1) return fail("truncated length");        // :190
        if (len > (1u << 20)) return fail("implausible token length");
        tokens[i].resize(len);
        if (len && fread(tokens[i].data(), 1, len, f) != len) return fail("truncated token"); // :192
        first_index_.emplace(tokens[i], (int)i);  // emplace keeps the FIRST index, like :209-213
    }
    fclose(f);
    return true;
}

int Tokenizer::lookup(std::string_view str) const
{
    auto it = first_index_.find(std::string(str));
    return it == first_index_.end() ? -1 : it->second;
}

static int utf8_len(unsigned char c)
{
    if (c < 0x80) return 1;
    if ((c & 0xE0) == 0xC0) return 2;
    if ((c & 0xF0) == 0xE0) return 3;
    if ((c & 0xF8) == 0xF0) return 4;
    return -1;  // std.unicode.utf8ByteSequenceLength error
}

// :236-245 one token per UTF-8 code point
bool Tokenizer::encode_code_points(std::string_view input, std::vector<int32_t> *out, std::string *err) const
{
    out->clear();
    if (max_token_len * 2 > 128) {  // :222-225 TokensTooLong
        if (err) *err = "TokensTooLong";
        return false;
    }
    size_t idx = 0;
    while (idx < input.size()) {
        const int n = utf8_len((unsigned char)input[idx]);
        if (n < 0 || idx + (size_t)n > input.size()) {
            if (err) *err = "Utf8InvalidStartByte";
            return false;
        }
        const int id = lookup(input.substr(idx, (size_t)n));
        if (id < 0) {
            if (err) *err = "TokenNotFound";  // :240-242
            return false;
        }
        out->push_back(id);
        idx += (size_t)n;
    }
    return true;
}

// The reference's merge loop as written (:247-278): every round scans ALL adjacent pairs for the
// best-scoring merge -- O(n^2) vocabulary lookups; a 2000-token prompt takes 1.3 s on a host core, five
// times the GPU's batched prefill of it.  Kept as the definition the fast form is tested against.
bool Tokenizer::encode_quadratic(std::string_view input, std::vector<int32_t> *out, std::string *err) const
{
    if (!encode_code_points(input, out, err)) return false;
    std::string cat;
    while (out->size() >= 2) {
        float best_score = -1e10f;  // :248
        int best_id = 0;
        long best_idx = -1;
        for (size_t i = 0; i + 1 < out->size(); i++) {
            cat.assign(tokens[(*out)[i]]);
            cat.append(tokens[(*out)[i + 1]]);
            const int id = lookup(cat);
            if (id >= 0 && scores[id] > best_score) {  // :261 strict: the earliest pair wins ties
                best_score = scores[id];
                best_id = id;
                best_idx = (long)i;
            }
        }
        if (best_idx < 0) break;  // :274-277
        (*out)[(size_t)best_idx] = best_id;
        out->erase(out->begin() + best_idx + 1);  // :272-273
    }
    return true;
}

// The same merges in the same order, O(n log n): every round of :247-278 picks, among the pairs that are
// adjacent NOW, the one with the highest score, the leftmost among equals.  A heap of candidate merges
// ordered by (score descending, position ascending) yields exactly that pair once stale entries -- pairs
// one of whose sides has been merged away since -- are skipped; after a merge only the two pairs around the
// new token are new.  The tokens stay in a linked list in their original order, so the position of a pair
// is the original index of its left token.
bool Tokenizer::encode(std::string_view input, std::vector<int32_t> *out, std::string *err) const
{
    if (!encode_code_points(input, out, err)) return false;
    const int n = (int)out->size();
    if (n < 2) return true;
    struct Cand {
        float score;
        int left, right, id;          // list nodes (original indices) and the merged token
        uint32_t lver, rver;          // versions of the two nodes when the candidate was made
    };
    auto worse = [](const Cand &a, const Cand &b) {  // max-heap: a sinks below b if ...
        return a.score < b.score || (a.score == b.score && a.left > b.left);
    };
    std::vector<Cand> heap;
    std::vector<int> prev((size_t)n), next((size_t)n);
    std::vector<uint32_t> ver((size_t)n, 0);
    std::vector<char> alive((size_t)n, 1);
    std::vector<int32_t> &tok = *out;
    std::string cat;
    auto consider = [&](int l, int r) {
        cat.assign(tokens[(size_t)tok[(size_t)l]]);
        cat.append(tokens[(size_t)tok[(size_t)r]]);
        const int id = lookup(cat);
        if (id >= 0 && scores[(size_t)id] > -1e10f) {  // :248, :261
            heap.push_back({scores[(size_t)id], l, r, id, ver[(size_t)l], ver[(size_t)r]});
            std::push_heap(heap.begin(), heap.end(), worse);
        }
    };
    for (int i = 0; i < n; i++) { prev[(size_t)i] = i - 1; next[(size_t)i] = i + 1 < n ? i + 1 : -1; }
    for (int i = 0; i + 1 < n; i++) consider(i, i + 1);
    while (!heap.empty()) {
        std::pop_heap(heap.begin(), heap.end(), worse);
        const Cand c = heap.back();
        heap.pop_back();
        if (!alive[(size_t)c.left] || !alive[(size_t)c.right] || ver[(size_t)c.left] != c.lver ||
            ver[(size_t)c.right] != c.rver || next[(size_t)c.left] != c.right)
            continue;  // stale
        tok[(size_t)c.left] = c.id;     // :272
        ver[(size_t)c.left]++;
        alive[(size_t)c.right] = 0;     // :273
        const int nn = next[(size_t)c.right];
        next[(size_t)c.left] = nn;
        if (nn >= 0) prev[(size_t)nn] = c.left;
        if (prev[(size_t)c.left] >= 0) consider(prev[(size_t)c.left], c.left);
        if (nn >= 0) consider(c.left, nn);
    }
    size_t w = 0;
    for (int i = 0; i < n; i++)
        if (alive[(size_t)i]) tok[w++] = tok[(size_t)i];
    tok.resize(w);
    return true;
}

// ---- PRNG: Xoshiro256++ / SplitMix64 as in Zig's std.Random ----
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

void Prng::seed_with(uint64_t seed)
{
    uint64_t sm = seed;
    for (int i = 0; i < 4; i++) {  // SplitMix64.next()
        sm += 0x9E3779B97F4A7C15ULL;
        uint64_t z = sm;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        s_[i] = z ^ (z >> 31);
    }
}

uint64_t Prng::next_u64()
{
    const uint64_t r = rotl(s_[0] + s_[3], 23) + s_[0];
    const uint64_t t = s_[1] << 17;
    s_[2] ^= s_[0];
    s_[3] ^= s_[1];
    s_[1] ^= s_[2];
    s_[0] ^= s_[3];
    s_[2] ^= t;
    s_[3] = rotl(s_[3], 45);
    return r;
}

float Prng::next_f32()
{
    // std.Random.float(f32): 23 mantissa bits, exponent from the count of leading zeros
    const uint64_t rnd = next_u64();
    int lz = rnd ? __builtin_clzll(rnd) : 64;
    if (lz >= 41) {
        const uint64_t r2 = next_u64();
        lz = 41 + (r2 ? __builtin_clzll(r2) : 64);
        if (lz == 41 + 64) {
            const uint32_t r3 = (uint32_t)next_u64() | 1u;
            lz += __builtin_clz(r3);
        }
    }
    const uint32_t mantissa = (uint32_t)rnd & 0x7FFFFFu;
    const uint32_t exponent = (uint32_t)(126 - lz) << 23;
    const uint32_t bits = exponent | mantissa;
    float f;
    std::memcpy(&f, &bits, sizeof f);
    return f;
}

void softmax(float *x, size_t n)
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

size_t argmax(const float *x, size_t n)
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

size_t sample(const float *probs, size_t n, Prng &rng)
{
    const float r = rng.next_f32();  // :731
    float cdf = 0.0f;
    for (size_t i = 0; i < n; i++) {
        cdf += probs[i];
        if (r < cdf) return i;
    }
    return n - 1;  // :740
}

// Descending by probability, ties to the lower token id -- the permutation std::sort gives with that
// (total) comparator, computed as a stable LSD radix sort on the bit patterns: the candidates are
// non-negative floats (>= the cutoff), whose bit patterns order like their values, and they arrive in
// ascending token id, which a stable sort keeps among equal keys.  With the reference's defaults
// (-t 1.0 -p 0.9) this sort runs once per generated token over up to 32000 candidates: 1.1 ms as a
// comparison sort, more than a whole stories110M forward pass (0.33 ms) -- the radix form takes ~0.1 ms.
static void sort_desc(std::vector<IndexedF32> &v)
{
    const size_t n = v.size();
    auto cmp = [](const IndexedF32 &a, const IndexedF32 &b) {
        return a.value > b.value || (a.value == b.value && a.index < b.index);
    };
    if (n < 1024) {
        std::sort(v.begin(), v.end(), cmp);
        return;
    }
    auto key = [](const IndexedF32 &e) {
        uint32_t b;
        memcpy(&b, &e.value, sizeof b);
        return ~b;  // ascending in ~bits = descending in value
    };
    static thread_local std::vector<IndexedF32> tmp;
    tmp.resize(n);
    constexpr int kBits = 11, kBuckets = 1 << kBits;
    std::vector<uint32_t> hist(3 * kBuckets, 0);
    for (const IndexedF32 &e : v) {
        const uint32_t k = key(e);
        hist[k & (kBuckets - 1)]++;
        hist[kBuckets + ((k >> kBits) & (kBuckets - 1))]++;
        hist[2 * kBuckets + (k >> (2 * kBits))]++;
    }
    for (int pass = 0; pass < 3; pass++) {
        uint32_t *h = hist.data() + pass * kBuckets, sum = 0;
        for (int b = 0; b < kBuckets; b++) {
            const uint32_t c = h[b];
            h[b] = sum;
            sum += c;
        }
        const int shift = pass * kBits;
        IndexedF32 *src = pass == 1 ? tmp.data() : v.data(), *dst = pass == 1 ? v.data() : tmp.data();
        for (size_t i = 0; i < n; i++) {
            const uint32_t b = (key(src[i]) >> shift) & (pass == 2 ? 0x3ffu : (uint32_t)(kBuckets - 1));
            dst[h[b]++] = src[i];
        }
    }
    v.swap(tmp);  // three passes: v -> tmp -> v -> tmp
}

size_t sample_top_p(const float *probs, size_t n, float p, std::vector<IndexedF32> &scratch,
                    Prng &rng)
{
    return sample_top_p_margin(probs, n, p, scratch, rng, nullptr);
}

size_t sample_top_p_margin(const float *probs, size_t n, float p, std::vector<IndexedF32> &scratch,
                           Prng &rng, float *margin)
{
    if (margin) *margin = 1.0f;
    // :759-770 candidates below (1-p)/(n-1) cannot be in the nucleus
    const float cutoff = (1.0f - p) / ((float)n - 1.0f);
    scratch.clear();
    for (size_t i = 0; i < n; i++)
        if (probs[i] >= cutoff) scratch.push_back({(uint32_t)i, probs[i]});
    if (scratch.empty()) return argmax(probs, n);  // reference asserts; be safe in release
    // :774 sorts descending by probability with std.sort.pdq, an UNSTABLE sort whose order among
    // equal probabilities is an implementation detail of Zig's library (not available here).  The
    // comparator is made total -- ties go to the lower token id -- so that the nucleus and the
    // sampled token are at least deterministic across standard libraries; with tied probabilities
    // at the cut they can differ from the reference binary's for the same seed.
    sort_desc(scratch);
    float cumulative = 0.0f;
    size_t cutoff_index = scratch.size() - 1;  // :778
    for (size_t i = 0; i < scratch.size(); i++) {
        cumulative += scratch[i].value;
        if (cumulative > p) {  // :781
            cutoff_index = i;
            break;
        }
    }
    const float r = rng.next_f32() * cumulative;  // :789
    float cdf = 0.0f;
    if (margin) {  // how close the draw is to a boundary of the truncated cdf (tests: is a flip a near tie?)
        float m = r;  // the boundary at 0
        for (size_t i = 0; i < cutoff_index; i++) {  // the last boundary is not one: everything beyond falls to :797
            cdf += scratch[i].value;
            m = std::fmin(m, std::fabs(r - cdf));
        }
        *margin = m;
        cdf = 0.0f;
    }
    for (size_t i = 0; i <= cutoff_index; i++) {
        cdf += scratch[i].value;
        if (r < cdf) return scratch[i].index;
    }
    return scratch[cutoff_index].index;  // :797
}

int is_raw_byte(std::string_view s)
{
    if (s.size() != 6) return -1;
    if (s[0] != '<' || s[1] != '0' || s[2] != 'x' || s[5] != '>') return -1;
    int byte = 0;
    for (int i = 3; i < 5; i++) {
        const char c = s[(size_t)i];
        byte *= 16;
        if (c >= '0' && c <= '9') byte += c - '0';
        else if (c >= 'a' && c <= 'f') byte += c - 'a' + 10;
        else if (c >= 'A' && c <= 'F') byte += c - 'A' + 10;
        else return -1;
    }
    // std.ascii.isPrint (0x20..0x7e) or isWhitespace (' ', \t \n \r \v \f)
    const bool print = byte >= 0x20 && byte <= 0x7e;
    const bool space = byte == ' ' || (byte >= 9 && byte <= 13);
    return (print || space) ? byte : -1;
}

}  // namespace l2zhost

// ---- C hooks so the tests can drive the host logic through ctypes ----
using namespace l2zhost;

extern "C" {

void *l2zh_tokenizer_open(const char *path, size_t vocab_size, char *err, size_t err_cap)
{
    auto *t = new Tokenizer();
    std::string e;
    if (!t->from_file(path, vocab_size, &e)) {
        if (err && err_cap) snprintf(err, err_cap, "%s", e.c_str());
        delete t;
        return nullptr;
    }
    return t;
}
void l2zh_tokenizer_close(void *t) { delete static_cast<Tokenizer *>(t); }
int l2zh_tokenizer_lookup(void *t, const char *bytes, size_t n)
{
    return static_cast<Tokenizer *>(t)->lookup(std::string_view(bytes, n));
}
uint32_t l2zh_tokenizer_max_token_len(void *t) { return static_cast<Tokenizer *>(t)->max_token_len; }
size_t l2zh_tokenizer_token(void *t, int id, char *out, size_t cap)
{
    const std::string &s = static_cast<Tokenizer *>(t)->tokens[(size_t)id];
    const size_t n = s.size() < cap ? s.size() : cap;
    std::memcpy(out, s.data(), n);
    return s.size();
}
// returns the token count, or -1 on error
long l2zh_tokenizer_encode(void *t, const char *bytes, size_t n, int32_t *out, size_t cap)
{
    std::vector<int32_t> v;
    std::string e;
    if (!static_cast<Tokenizer *>(t)->encode(std::string_view(bytes, n), &v, &e)) return -1;
    for (size_t i = 0; i < v.size() && i < cap; i++) out[i] = v[i];
    return (long)v.size();
}
// the reference's merge loop as written, for tests of the fast form
long l2zh_tokenizer_encode_quadratic(void *t, const char *bytes, size_t n, int32_t *out, size_t cap)
{
    std::vector<int32_t> v;
    std::string e;
    if (!static_cast<Tokenizer *>(t)->encode_quadratic(std::string_view(bytes, n), &v, &e)) return -1;
    for (size_t i = 0; i < v.size() && i < cap; i++) out[i] = v[i];
    return (long)v.size();
}
int l2zh_is_raw_byte(const char *s, size_t n) { return is_raw_byte(std::string_view(s, n)); }
void l2zh_prng_floats(uint64_t seed, float *out, size_t n)
{
    Prng r(seed);
    for (size_t i = 0; i < n; i++) out[i] = r.next_f32();
}
uint64_t l2zh_prng_u64(uint64_t seed, size_t skip)
{
    Prng r(seed);
    for (size_t i = 0; i < skip; i++) r.next_u64();
    return r.next_u64();
}
size_t l2zh_sample(const float *probs, size_t n, uint64_t seed)
{
    Prng r(seed);
    return sample(probs, n, r);
}
size_t l2zh_sample_top_p(const float *probs, size_t n, float p, uint64_t seed)
{
    Prng r(seed);
    std::vector<IndexedF32> scratch;
    return sample_top_p(probs, n, p, scratch, r);
}
void l2zh_softmax(float *x, size_t n) { softmax(x, n); }
// one generator across calls, as the generation loop uses it (main.zig:845 / :926, then :1009-1012 per position)
void *l2zh_prng_open(uint64_t seed) { return new Prng(seed); }
void l2zh_prng_close(void *rng) { delete static_cast<Prng *>(rng); }
size_t l2zh_sample_top_p_rng(const float *probs, size_t n, float p, void *rng, float *margin)
{
    std::vector<IndexedF32> scratch;
    return sample_top_p_margin(probs, n, p, scratch, *static_cast<Prng *>(rng), margin);
}
size_t l2zh_sample_rng(const float *probs, size_t n, void *rng) { return sample(probs, n, *static_cast<Prng *>(rng)); }
}
