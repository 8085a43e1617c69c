// tunables.h -- every environment knob of the library, read ONCE (first use) in one place.
// None of them changes results except where noted; they exist for A/B measurements on the GPU box
// (scripts/) and for the tests that put several ranks on one GPU.  Launch paths only
// ever see this struct.
#pragma once

namespace l2z {

struct Tunables {
    // --- decode mat-vec (matvec.hip) ---
    int row_kernel = 1;        // L2Z_ROW_KERNEL      0: wide rows take the per-wave kernel too
    int row_blocks = 2;        // L2Z_ROW_BLOCKS      resident row-kernel blocks per CU
    int row_tail_skip = 1;     // L2Z_ROW_TAIL_SKIP   0: a wide row's last, partly filled batch re-reads the row start instead of skipping the loads
    int max_blocks_per_cu = 8; // L2Z_MAX_BLOCKS_PER_CU
    int grid_cap = 0;          // L2Z_GRID_CAP        max blocks of one mat-vec launch (0: none); set when
                               //                     several ranks share one GPU so that a kernel
                               //                     waiting for a peer leaves the peer room to run
    // --- decode attention (attention.hip, runstate.cpp) ---
    int attn_block = 0;        // L2Z_ATTN_BLOCK      force the one-block-per-head kernel's block size
    int attn_split = -1;       // L2Z_ATTN_SPLIT      0: never split; n > 0: n chunks at every position
                               //                     (changes rounding: chunk count is part of the arithmetic)
    int attn_split_pos = -1;   // L2Z_ATTN_SPLIT_POS  first position that uses the split form (default 256)
    int attn_split_wide_pos = -1;  // L2Z_ATTN_SPLIT_WIDE_POS  first position at which the split form runs 1024 threads per block (256 below; default 1024)
    int attn_short_pos = -1;   // L2Z_ATTN_SHORT_POS  positions below this take the 256-thread one-block-per-head kernel with the
                               //                     speculative first round whatever seq_len is (default: by head size, 0: never)
    int attn_pos_arg = 0;      // L2Z_ATTN_POS_ARG    experiment (l2z_time_kind only): the split kernel gets pos by value, not from device memory
    int fuse_small = 1;        // L2Z_FUSE_SMALL      0: small models keep separate qkv / attention launches
    // --- graphs / transport (runstate.cpp, comm.cpp, forward.cpp) ---
    int no_graph = 0;          // L2Z_NO_GRAPH        1: launch eagerly
    int comm_graph = 1;        // L2Z_COMM_GRAPH      0: RCCL collectives are launched eagerly, not captured
    int prefer_rccl = 0;       // L2Z_COMM=rccl       use RCCL even when the peer-write transport is connected
    int p2p_push = 1;          // L2Z_P2P_PUSH        1: producers push their outputs from their epilogues where consumers read the words
                               //                     (consumer-side form); 2: also where a gather / reduce launch
                               //                     collects them (slower, measured); 0: never (no consumer-side form then)
    int p2p_consume = -1;      // L2Z_P2P_CONSUME     1: consumers read their gathered input as LL words while staging x (no gather launches);
                               //                     0: a gather launch per gathered vector (consumers read plain buffers); -1 (default):
                               //                     by shape -- the consumer-side form for up to 2 ranks or rows narrower than 4096, gather
                               //                     launches beyond (one rank of N alone, 7B shape, profiles/r04_solo_rank.md: N = 2 equal,
                               //                     N = 4 +13 %, N = 8 +27 % for the gather launches)
    int argmax_xchg = 1;       // L2Z_ARGMAX_XCHG     0: greedy steps of a shard group gather all the logits and scan them (round 4's form)
                               //                     instead of exchanging one (max, first index) candidate per rank (peer-write transport)
    int reduce_block = 128;    // L2Z_REDUCE_BLOCK    threads per block of scheme B's reduce launch (64 ... 1024; one element per thread;
                               //                     one rank of 8 alone: 64 / 128 / 256 / 512 / 1024 threads -> 734 / 733 / 728 / 722 / 693 tok/s)
    long long p2p_timeout_s = 20;  // L2Z_P2P_TIMEOUT_S
    int scheme_b = 0;          // L2Z_SCHEME_B        1: shard groups take scheme B: Wo / W2 sharded by COLUMNS, every rank's partial
                               //                     [dim] vectors summed by an all-reduce -- 2 collectives per layer instead of 4 all-gathers, but the
                               //                     sum order differs from the unsharded pass (logit tolerance, not bit identity).  Read when Weights /
                               //                     RunState objects are created
    int p2p_bulk_mb = -1;      // L2Z_P2P_BULK_MB     MB per bulk landing region of the peer-write arena (two of
                               //                     them; sharded prefill); default: longest vector x chunk tokens
    // --- batched prefill (prefill_host.cpp, prefill_*.hip) ---
    int prefill = 1;           // L2Z_PREFILL         0: prompts are stepped token by token
    int pf_chunk = 0;          // L2Z_PF_CHUNK        tokens per chunk, fixed (0: 1024 while that many remain, then 512, then the rest)
    int pf_skinny_form = 1;    // L2Z_PF_SKINNY_FORM  short-prompt GEMM: 1 LDS-staged (direct-to-LDS ring where K % 256 == 0), 2 register-staged LDS form only, 0 no LDS
    int pf_tile = 0;           // L2Z_PF_TILE         force a tile form of the prefill GEMM (experiments: 2 64x64, 8 128x64, 9 32x64, 10 32x32, 11 128x128;
                               //                     disables the paired / fused launches); 0: chosen by grid fill
    int pf_skinny_spread = 1;  // L2Z_PF_SKINNY_SPREAD 0: the short-prompt kernels' blocks take the feature groups in order, not one window of rows per XCD
    int pf_skinny_max = -1;    // L2Z_PF_SKINNY_MAX   longest chunk that takes the short-prompt GEMMs (default 64 tokens)
    int pf_skinny_tms = 0;     // L2Z_PF_SKINNY_TMS   token tiles (of 16) per block of the short-prompt GEMM: 1, 2, 4 (register form); 0: by prompt length
    int pf_attn = 1;           // L2Z_PF_ATTN         0: per-query prefill attention only; 2: the LDS-softmax tiled kernel instead of the flash form; 3: flash form with one key part (4 waves)
    int pf_dma = 1;            // L2Z_PF_DMA          0: GEMM operands staged through registers instead of direct-to-LDS loads
    int pf_order = 1;          // L2Z_PF_ORDER        0: 2-D grids for the tile GEMM (x = feature tile, y = token tile)
    int pf_fuse = 1;           // L2Z_PF_FUSE         0: separate Q / K / V and W1 / W3 GEMMs
    int pf_kgs = -1;           // L2Z_PF_KGS          the tile GEMM's two k-groups on two blocks (same bits, twice the blocks): -1 by grid fill,
                               //                     0 never, 10 + f: always, on tile form f (0 128x64, 1 64x64, 2 32x64, 4 128x128)
    int pf_panel = 1;          // L2Z_PF_PANEL        0: chunks of 17 ... 96 tokens keep the short-prompt / tile GEMMs instead of the
                               //                     K-range panel kernel (prefill_panel.hip; changes rounding: the ranges are part of the arithmetic)
    int pf_panel_form = 0;     // L2Z_PF_PANEL_FORM   9: round 5's forms at three / four token tiles (ranges of 256, three ring buffers; changes rounding), for A/B
    int pf_panel_max = -1;     // L2Z_PF_PANEL_MAX    longest chunk that takes the panel kernel (default and maximum 96 tokens)
    int pf_panel_min = -1;     // L2Z_PF_PANEL_MIN    shortest chunk that takes it (default 17: up to 16 tokens the short-prompt GEMMs are ahead)
    int pf_splitk = -1;        // L2Z_PF_SPLITK       K ranges per output tile of the tile GEMM for chunks of <= 256 tokens: -1 by shape,
                               //                     1 none, 2 / 4 forced (changes rounding: the range partials are added in range order)
};

const Tunables &tunables();
int prefill_chunk_tokens();          // the longest chunk of a batched prefill: L2Z_PF_CHUNK clamped to [16, 2048], default 1024
int prefill_next_chunk(int remaining);  // tokens of the next chunk (default: 1024 while that many remain, then 512, then the rest)
// Override one knob by its environment name after start-up (measurement harnesses that try several
// settings in one process: include/llama2_hip_test.h l2z_option_set).  False: unknown name.
bool tunables_set(const char *env_name, long long value);

}  // namespace l2z
