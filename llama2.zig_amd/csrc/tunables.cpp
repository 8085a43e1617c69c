// tunables.cpp -- the one place the environment is read (see tunables.h).
#include "tunables.h"

#include <cstdlib>
#include <cstring>

namespace l2z {
namespace {

void env_int(const char *name, int *v)
{
    if (const char *e = getenv(name))
        if (*e) *v = atoi(e);
}

Tunables read_env()
{
    Tunables t;
    env_int("L2Z_GRID_CAP", &t.grid_cap);
    env_int("L2Z_ATTN_SPLIT", &t.attn_split);
    env_int("L2Z_ATTN_SPLIT_POS", &t.attn_split_pos);
    env_int("L2Z_FUSE_SMALL", &t.fuse_small);
    env_int("L2Z_SCHEME_B", &t.scheme_b);
    env_int("L2Z_NO_GRAPH", &t.no_graph);
    if (const char *e = getenv("L2Z_COMM")) t.prefer_rccl = strcmp(e, "rccl") == 0;
    env_int("L2Z_ARGMAX_XCHG", &t.argmax_xchg);
    env_int("L2Z_P2P_CONSUME", &t.p2p_consume);
    if (const char *e = getenv("L2Z_P2P_TIMEOUT_S"))
        if (*e) t.p2p_timeout_s = atoll(e);
    env_int("L2Z_P2P_BULK_MB", &t.p2p_bulk_mb);
    env_int("L2Z_PREFILL", &t.prefill);
    env_int("L2Z_PF_CHUNK", &t.pf_chunk);
    env_int("L2Z_PF_PANEL", &t.pf_panel);
    env_int("L2Z_PF_PANEL_MAX", &t.pf_panel_max);
    env_int("L2Z_PF_X3", &t.pf_x3);
    env_int("L2Z_PF_X3_STREAM_MIN", &t.pf_x3_stream_min);
    env_int("L2Z_PF_FUSE_PLANES", &t.pf_fuse_planes);
    return t;
}

}  // namespace

namespace {
Tunables &mutable_tunables()
{
    static Tunables t = read_env();
    return t;
}
}  // namespace

const Tunables &tunables() { return mutable_tunables(); }

int prefill_chunk_tokens()
{
    int n = tunables().pf_chunk > 0 ? tunables().pf_chunk : 1024;
    if (n < 16) n = 16;
    if (n > 2048) n = 2048;
    return n;
}

int prefill_next_chunk(int remaining)
{
    if (tunables().pf_chunk > 0) return remaining < prefill_chunk_tokens() ? remaining : prefill_chunk_tokens();
    // 1024 tokens at a time while that many are left (128 x 128 tiles: one block per CU for N = 4096), else
    // 512 (128 x 64 tiles), else the rest -- a chunk of 513 .. 1023 tokens would leave a wave of tiles
    // mostly empty
    return remaining >= 1024 ? 1024 : remaining < 512 ? remaining : 512;
}

bool tunables_set(const char *name, long long v)
{
    Tunables &t = mutable_tunables();
    struct { const char *n; int *p; } ints[] = {
        {"L2Z_GRID_CAP", &t.grid_cap}, {"L2Z_ATTN_SPLIT", &t.attn_split}, {"L2Z_ATTN_SPLIT_POS", &t.attn_split_pos},
        {"L2Z_FUSE_SMALL", &t.fuse_small}, {"L2Z_NO_GRAPH", &t.no_graph}, {"L2Z_SCHEME_B", &t.scheme_b},
        {"L2Z_COMM_RCCL", &t.prefer_rccl}, {"L2Z_ARGMAX_XCHG", &t.argmax_xchg}, {"L2Z_P2P_CONSUME", &t.p2p_consume},
        {"L2Z_P2P_BULK_MB", &t.p2p_bulk_mb}, {"L2Z_PREFILL", &t.prefill}, {"L2Z_PF_CHUNK", &t.pf_chunk},
        {"L2Z_PF_PANEL", &t.pf_panel}, {"L2Z_PF_PANEL_MAX", &t.pf_panel_max},
        {"L2Z_PF_X3", &t.pf_x3}, {"L2Z_PF_X3_STREAM_MIN", &t.pf_x3_stream_min},
        {"L2Z_PF_FUSE_PLANES", &t.pf_fuse_planes}};
    for (auto &e : ints)
        if (strcmp(e.n, name) == 0) {
            *e.p = (int)v;
            return true;
        }
    if (strcmp(name, "L2Z_P2P_TIMEOUT_S") == 0) {
        t.p2p_timeout_s = v;
        return true;
    }
    return false;
}

}  // namespace l2z
