// tunables.h -- every environment knob of the library, read ONCE (first use) in one place.
// None of them changes results except where noted.  Round 6 cut the list from 39 to 15: a knob stays if a test, a bench
// leg or a deployment needs it (its line says which, and what the measured effect is); settings that only ever LOST an
// A/B are constants now (second block: the value, and the measurement that fixed it) and the forms they selected are gone.
// Launch paths only ever see this struct.
#pragma once

namespace l2z {

struct Tunables {
    // ---- knobs (environment variable of the same name in capitals, or l2z_option_set in tests) ----
    int grid_cap = 0;          // L2Z_GRID_CAP        max blocks of one mat-vec launch (0: none).  Needed when several ranks share ONE GPU
                               //                     (tests, the one-GPU proxy of bench --gpus N): a launch that polls for a peer's words must
                               //                     leave the peer's kernels room to run; never set with a GPU per rank
    int attn_split = -1;       // L2Z_ATTN_SPLIT      0: never split the decode attention; n > 0: n chunks per head at every position (changes
                               //                     rounding: the chunk count is part of the arithmetic).  Tests drive the split form on toy
                               //                     contexts with it; default min(8, CUs / heads): 19.7 us per layer at pos 2047 of the 7B shape
    int attn_split_pos = -1;   // L2Z_ATTN_SPLIT_POS  first position that takes the split form (default 256: below, one block per head wins by 1-3 us)
    int fuse_small = 1;        // L2Z_FUSE_SMALL      0: small MHA models keep separate qkv / attention launches (fused: 5 -> 4 launches per layer;
                               //                     rounds differently from the unfused launches, so tests that compare with a shard group set 0)
    int no_graph = 0;          // L2Z_NO_GRAPH        1: launch eagerly (debugging, per-launch PMC passes; a graph replay is one host call per token)
    int prefer_rccl = 0;       // L2Z_COMM=rccl       use RCCL even when the peer-write transport is connected (bench's rccl legs)
    int p2p_consume = -1;      // L2Z_P2P_CONSUME     1: consumers read their gathered input as LL words while staging x (no gather launches);
                               //                     0: a gather launch per gathered vector; -1 (default): by shape -- the consumer-side form for
                               //                     up to 2 ranks or rows narrower than 4096, gather launches beyond (one rank of N alone, 7B:
                               //                     N = 2 equal, N = 4 +13 %, N = 8 +27 % for the gather launches; bench's p2p legs set it)
    int argmax_xchg = 1;       // L2Z_ARGMAX_XCHG     0: greedy steps of a shard group gather all the logits and scan them instead of exchanging one
                               //                     (max, first index) candidate per rank (+2-3 % on the solo bed); 0 also keeps l2z_logits_read
                               //                     & co. non-collective after greedy steps
    long long p2p_timeout_s = 20;  // L2Z_P2P_TIMEOUT_S  seconds a peer-write wait spins before it latches L2Z_ERR_COMM
    int scheme_b = 0;          // L2Z_SCHEME_B        1: shard groups take scheme B: Wo / W2 sharded by COLUMNS, every rank's partial
                               //                     [dim] vectors summed by an all-reduce -- 2 collectives per layer instead of 4 all-gathers (+12 %
                               //                     at N = 8 on the solo bed), but the sum order differs from the unsharded pass (logit tolerance,
                               //                     not bit identity).  Read when Weights / RunState objects are created
    int p2p_bulk_mb = -1;      // L2Z_P2P_BULK_MB     MB per bulk landing region of the peer-write arena (two of them; sharded prefill); default:
                               //                     widest matrix x chunk tokens (90 MB at 7B); 0: none (sharded prompts are stepped)
    int prefill = 1;           // L2Z_PREFILL         0: prompts are stepped token by token (the reference's own order of operations)
    int pf_chunk = 0;          // L2Z_PF_CHUNK        tokens per prefill chunk, fixed (0: 1024 while that many remain, then 512, then the rest);
                               //                     bounds the prefill scratch (tests; hosts short of memory)
    int pf_panel = 1;          // L2Z_PF_PANEL        0: chunks of 17 ... 96 tokens keep the short-prompt / tile GEMMs instead of the K-range panel
                               //                     kernel (changes rounding: the ranges are part of the arithmetic); panel: 32 / 64 / 96 tokens
                               //                     8.8 / 12.5 / 17.6 -> 6.6 / 10.5 / 14.6 ms at 7B
    int pf_x3 = 1;             // L2Z_PF_X3           0: every GEMM of the batched prefill multiplies on the f32 matrix cores (v_mfma_f32_32x32x2_f32:
                               //                     the arithmetic of rounds 2-5; the A/B of the accuracy tests); 1 (default): matrices that stream
                               //                     from HBM multiply on the bf16 cores over three-term splits of both operands (six
                               //                     v_mfma_f32_32x32x16_bf16 per 16 k; changes rounding, error against float64 not above the f32
                               //                     chain's): 7B 128 / 512 / 1024 tokens 17.7 / 57.4 / 108 -> 15.3 / 40.6 / 74 ms; 2: every matrix
                               //                     (tests drive the planes kernels on toy shapes)
    int pf_x3_stream_min = 33; // L2Z_PF_X3_STREAM_MIN shortest chunk that takes the STREAM form of the bf16 kernel (up to 128 tokens; below, the
                               //                     panel kernel): tests of both sides of the switch-over (7B, ms: 32 tokens 6.6 vs 7.3,
                               //                     36 / 40 / 48 tokens 8.6 / 8.7 / 8.8 vs 8.0 / 8.2 / 8.2: profiles/r06t)
    int pf_fuse_planes = 1;    // L2Z_PF_FUSE_PLANES  0: every GEMM on the bf16 cores splits its activation matrix in a launch of its own just
                               //                     before it, and Wo / W2 add their K ranges' sums themselves (same bits; the A/B of the
                               //                     unsharded pass's cross-kernel work: planes written by rmsnorm, the attention output, the
                               //                     SwiGLU epilogue -- four launches per layer -- and the ranges' sums of Wo / W2 added by
                               //                     the rmsnorm launch behind them)
    int pf_panel_max = -1;     // L2Z_PF_PANEL_MAX    longest chunk that takes the panel kernel (default and maximum 96 tokens): tests of both sides
                               //                     of the switch-over

    // ---- settled by measurement: constants (the knob is gone; profiles/ has the A/B that fixed each) ----
    static constexpr int row_blocks = 2;         // resident row-kernel blocks per CU: 219 tok/s; 1 -> 208, 3 -> 213, 4-8 -> 208-211 (r01)
    static constexpr int max_blocks_per_cu = 8;  // narrow-row mat-vec: occupancy cap
    static constexpr int p2p_push = 1;           // producers push from their epilogues only where consumers read the words (pushing where a
                                                 // gather / reduce launch collects anyway: 727 -> 580 tok/s on the solo bed, r04_solo_rank.md)
    static constexpr int reduce_block = 128;     // threads per block of scheme B's reduce launch (64 ... 1024: 734 / 733 / 728 / 722 / 693 tok/s)
    static constexpr int pf_skinny_max = 64;     // longest chunk the short-prompt GEMMs take when the panel kernel does not apply
};

const Tunables &tunables();
int prefill_chunk_tokens();          // the longest chunk of a batched prefill: L2Z_PF_CHUNK clamped to [16, 2048], default 1024
int prefill_next_chunk(int remaining);  // tokens of the next chunk (default: 1024 while that many remain, then 512, then the rest)
// Override one knob by its environment name after start-up (measurement harnesses that try several
// settings in one process: include/llama2_hip_test.h l2z_option_set).  False: unknown name.
bool tunables_set(const char *env_name, long long value);

}  // namespace l2z
