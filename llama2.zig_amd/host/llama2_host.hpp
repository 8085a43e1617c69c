// llama2_host.hpp -- host side of the generation loop, mirroring the parts of
// cgbur/llama2.zig's src/main.zig that sit ABOVE transformer(): tokenizer
// (:166-283), samplers (:728-798), raw-byte token formatting (:1055-1076).
// SURVEY.md section 8(f) rows 1-2 ("next" after the hot path).  Plain C++17, no HIP.
#pragma once

#include <cstdint>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

namespace l2zhost {

// src/main.zig:166-283
class Tokenizer {
public:
    std::vector<std::string> tokens;
    std::vector<float> scores;
    uint32_t max_token_len = 0;

    // :173-196  file = u32 max_token_len, then vocab_size x {f32 score, u32 len, bytes}
    bool from_file(const std::string &path, size_t vocab_size, std::string *err);
    // :208-215  first index whose bytes equal str, or -1
    int lookup(std::string_view str) const;
    // :219-282  UTF-8 code points -> tokens, then greedy best-score pair merges
    bool encode(std::string_view input, std::vector<int32_t> *out, std::string *err) const;
    // the merge loop exactly as the reference writes it (O(n^2)); encode() gives the same tokens
    bool encode_quadratic(std::string_view input, std::vector<int32_t> *out, std::string *err) const;
    bool encode_code_points(std::string_view input, std::vector<int32_t> *out, std::string *err) const;

private:
    std::unordered_map<std::string, int> first_index_;  // same answer as the linear scan
};

// std.Random.DefaultPrng = Xoshiro256++ seeded through SplitMix64 (Zig std, restated
// from the published algorithm; the Zig source is not in the build image, so the exact
// stream is "parity unpinned").  main.zig:815, :845, :926.
class Prng {
public:
    explicit Prng(uint64_t seed = 0) { seed_with(seed); }
    void seed_with(uint64_t seed);
    uint64_t next_u64();
    float next_f32();  // std.Random.float(f32): uniform in [0,1)

private:
    uint64_t s_[4];
};

// src/main.zig:687-706 (the sampler re-uses softmax on the host-side logits, :1008)
void softmax(float *x, size_t n);
// :715-726
size_t argmax(const float *x, size_t n);
// :728-741
size_t sample(const float *probs, size_t n, Prng &rng);
// :752-798
struct IndexedF32 {
    uint32_t index;
    float value;
};
size_t sample_top_p(const float *probs, size_t n, float p, std::vector<IndexedF32> &scratch,
                    Prng &rng);
// the same draw; *margin (if not null) = distance of the scaled coin to the nearest inner boundary of the
// truncated cumulative distribution: how close this draw was to picking a neighbouring candidate
size_t sample_top_p_margin(const float *probs, size_t n, float p, std::vector<IndexedF32> &scratch,
                           Prng &rng, float *margin);
// :1055-1076  "<0xXX>" -> byte, only if printable or whitespace; -1 otherwise
int is_raw_byte(std::string_view s);

}  // namespace l2zhost
