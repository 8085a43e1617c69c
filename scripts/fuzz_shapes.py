"""Random small model shapes through the C ABI against the CPU oracle: decode (stepped + device
greedy loop), batched prefill, split attention.  usage: fuzz_shapes.py [n_configs] [seed]
tests/test_gpu_fuzz.py runs a seeded batch of it."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, __graft_entry__ as ge


def check_config(B, ck, orc, rng, cfg, shared, seed, log=print):
    """One shape: returns True if everything is within tolerance."""
    dim, hidden, hs, vocab, seq = cfg.dim, cfg.hidden_dim, cfg.dim // cfg.n_heads, cfg.vocab_size, cfg.seq_len
    blob = ck.synth_blob(cfg, shared, seed=seed)
    w = B.Weights(cfg, blob, shared); s = B.RunState(cfg); m = orc.Model(cfg.as_i32(), blob, shared)
    n_pos = min(seq, 40 if seq < 300 else 290)
    toks = [1] + rng.integers(0, vocab, n_pos - 1).tolist()
    worst = 0.0
    for pos, t in enumerate(toks):
        ref = m.transformer(t, pos); s.transformer(t, pos, w)
        if pos in (0, 1, n_pos // 2, n_pos - 1):
            got = s.logits()
            worst = max(worst, float(np.abs(got - ref).max() / (1e-3 + np.abs(ref).max())))
    ok = worst < 2e-4
    # device greedy loop vs oracle loop (first divergence must be a near tie)
    ref_t, marg = m.generate_greedy(toks[1:6], min(seq, 30))
    s.greedy_begin(toks[1:6]); dev = s.greedy_run(w, min(seq, 30))
    n = min(len(dev), len(ref_t)); same = next((i for i in range(n) if dev[i] != ref_t[i]), n)
    ok = ok and (same == n or marg[same] < 1e-4)
    # prefill vs stepped (when the shape allows it)
    pf = "-"
    if dim % 4 == 0 and hidden % 4 == 0 and hs % 4 == 0:
        s2 = B.RunState(cfg)
        s2.prefill(toks, 0, w)
        for pos, t in enumerate(toks):
            s.transformer(t, pos, w)
        d = float(np.abs(s2.logits() - s.logits()).max() / (1e-3 + np.abs(s.logits()).max()))
        pf = f"{d:.1e}"; ok = ok and d < 2e-4
        s2.close()
    log(f"{'ok ' if ok else 'BAD'} dim {dim} hs {hs} H {cfg.n_heads} kv {cfg.n_kv_heads} hid {hidden} V {vocab} S {seq} "
        f"L {cfg.n_layers} shared {int(shared)}: rel {worst:.1e} greedy {same}/{n} prefill {pf}")
    s.close(); w.close(); m.close()
    return ok


def random_config(ck, rng):
    hs = int(rng.choice([2, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128]))
    n_kv = int(rng.choice([1, 2, 3, 4]))
    kv_mul = int(rng.choice([1, 1, 2, 3, 4]))
    n_heads = n_kv * kv_mul
    dim = hs * n_heads
    hidden = int(rng.choice([dim + 2, 2 * dim, 3 * dim + 4, 172, 768, 1024 + 4 * int(rng.integers(0, 64))]))
    hidden += hidden % 2
    vocab = int(rng.choice([37, 256, 1000, 4099]))
    seq = int(rng.choice([24, 96, 300]))
    L = int(rng.integers(1, 4))
    return ck.Config(dim, hidden, L, n_heads, n_kv, vocab, seq), bool(rng.integers(0, 2))


def random_wide_config(ck, rng):
    """Row-kernel and 64-lane widths: dim 1024..8192, head sizes 64..256, hidden up to 11008."""
    hs = int(rng.choice([64, 128, 256]))
    dim = int(rng.choice([1024, 1536, 2048, 2304, 3072, 4096, 4352, 5120, 8192]))
    if dim % hs:
        hs = 64
    n_heads = dim // hs
    divs = [d for d in (1, 2, 4, 8) if n_heads % d == 0]
    n_kv = n_heads // int(rng.choice(divs))
    hidden = int(rng.choice([dim, 2 * dim + 256, 11008, 4096 + 4 * int(rng.integers(0, 512))]))
    vocab = int(rng.choice([100, 4000]))
    seq = int(rng.choice([24, 300]))
    return ck.Config(dim, hidden, 1, n_heads, n_kv, vocab, seq), bool(rng.integers(0, 2))


def run(n_cfg, seed, log=print, wide=False):
    pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
    orc = ge.load_oracle()
    rng = np.random.default_rng(seed)
    bad = 0
    for it in range(n_cfg):
        cfg, shared = random_wide_config(ck, rng) if wide else random_config(ck, rng)
        try:
            bad += not check_config(B, ck, orc, rng, cfg, shared, 1000 + it, log)
        except Exception as e:  # noqa: BLE001
            log(f"ERR {cfg}: {e}")
            bad += 1
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    print("bad:", run(n, sd, wide=len(sys.argv) > 3 and sys.argv[3] == "wide"))
