"""Random shapes x prompt lengths x world sizes for the batched prefill (one process, one GPU):
  * l2z_prefill vs the stepped loop: last-position logits within the logit tolerance, KV cache within 2e-4;
  * row-sharded prefill on emulated ranks vs the unsharded prefill: logits and every rank's KV shard
    BIT-IDENTICAL (shapes whose row shards the batched path does not take are reported as refused).
  * "wide": matrices that stream from HBM (dim 1024 ... 3072, hidden_dim multiples of 128) and chunks of 9 ... 80 tokens
    -- the K-range panel kernel of prefill_panel.hip on every side of its switch-overs (16 | 17, 32 | 33, 48 | 49,
    64 | 65), also under scheme B (column-sharded Wo / W2: ranks bit-identical to each other, logits at the tolerance).
usage: fuzz_prefill.py [n_configs] [seed] [wide]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, __graft_entry__ as ge

TOL = 5e-5


def run(n_cfg, seed, log=print, wide=False):
    pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
    rng = np.random.default_rng(seed)
    bad = 0
    for it in range(n_cfg):
        world = int(rng.choice([1, 2, 4, 8]))
        hs = int(rng.choice([8, 16, 32, 64, 128]))
        n_kv = world * int(rng.choice([1, 2]))
        n_heads = n_kv * int(rng.choice([1, 2, 4]))
        dim = hs * n_heads
        if dim > 2048:
            continue
        hidden = world * 4 * int(rng.integers(8, 300))
        if rng.integers(0, 2):  # K % 256 == 0: the direct-to-LDS short-prompt form and whole tile stages
            hidden = max(256, hidden // 256 * 256)
        vocab = world * int(rng.integers(8, 300))
        n_tok = int(rng.choice([1, 3, 9, 16, 17, 31, 40, 64, 65, 100, 129, 200, 257, 300, 530, 1030, 1600]))
        scheme_b = False
        if wide:
            hs = int(rng.choice([64, 128]))
            dim = int(rng.choice([1024, 2048, 3072]))
            n_heads = dim // hs
            n_kv = n_heads // int(rng.choice([1, 2, 4]))
            if n_kv % world:
                n_kv = n_heads
            hidden = 128 * world * int(rng.integers(2048 // (128 * world) + 1, 9216 // (128 * world) + 1))
            n_tok = int(rng.choice([9, 16, 17, 20, 31, 32, 33, 40, 48, 49, 63, 64, 65, 72, 80, 81, 96, 97]))
            scheme_b = world > 1 and bool(rng.integers(0, 3) == 0)
        seq = n_tok + 8
        cfg = ck.Config(dim, hidden, int(rng.integers(1, 3)), n_heads, n_kv, vocab, seq)
        toks = [1] + rng.integers(2, vocab, n_tok - 1).tolist()
        tag = f"world {world}{' scheme B' if scheme_b else ''} dim {dim} hs {hs} H {n_heads} kv {n_kv} hid {hidden} V {vocab} L {cfg.n_layers} tokens {n_tok}"
        try:
            w0 = B.Weights(cfg, None, False, seed=70 + it)
            s0, s1 = B.RunState(cfg), B.RunState(cfg)
            for pos, t in enumerate(toks):
                s1.transformer(t, pos, w0)
            s0.prefill(toks, 0, w0)
            d = float(np.abs(s0.logits() - s1.logits()).max())
            ok = d <= TOL * (1 + float(np.abs(s1.logits()).max())) and bool(np.isfinite(s0.logits()).all())
            kvd, S = cfg.kv_dim, cfg.seq_len
            kd = max(float(np.abs(s0.read(nm, l * S * kvd, n_tok * kvd) - s1.read(nm, l * S * kvd, n_tok * kvd)).max())
                     for l in range(cfg.n_layers) for nm in ("key_cache", "value_cache"))
            ok = ok and kd <= 2e-4
            note = f"max |dlogit| {d:.2e} max |dKV| {kd:.2e}"
            if world > 1:
                B.option_set("L2Z_SCHEME_B", 1 if scheme_b else 0)
                comms = [B.Comm(r, world, None, 0, emulated=True) for r in range(world)]
                ws = [B.Weights(cfg, None, False, seed=70 + it, comm=c) for c in comms]
                ss = [B.RunState(cfg, comm=c) for c in comms]
                B.option_set("L2Z_SCHEME_B", 0)
                try:
                    B.emu_prefill(ss, ws, toks, 0)
                    if scheme_b:  # the products are split across ranks: tolerance against the unsharded pass, ranks identical
                        db = float(np.abs(ss[0].logits() - s0.logits()).max())
                        same = all(np.array_equal(x.logits(), ss[0].logits()) for x in ss) and db <= TOL * (1 + float(np.abs(s0.logits()).max()))
                        note += f", scheme B max |dlogit| {db:.2e} ranks identical: {same}"
                        ok = ok and same
                        raise StopIteration
                    same = all(np.array_equal(x.logits(), s0.logits()) for x in ss)
                    kvl = kvd // world
                    for l in range(cfg.n_layers):
                        for nm in ("key_cache", "value_cache"):
                            full = s0.read(nm, l * S * kvd, n_tok * kvd).reshape(n_tok, kvd)
                            for r in range(world):
                                mine = ss[r].read(nm, l * S * kvl, n_tok * kvl).reshape(n_tok, kvl)
                                same = same and np.array_equal(mine, full[:, r * kvl:(r + 1) * kvl])
                    ok = ok and same
                    note += f", sharded bit-identical: {same}"
                except StopIteration:
                    pass
                except B.L2ZError as e:
                    note += f", sharded refused ({e.code})"
                    ok = ok and e.code == B.ERR_INVALID
                for o in ss + ws:
                    o.close()
                for c in comms:
                    c.close()
            log(f"{'ok ' if ok else 'BAD'} {tag}: {note}")
            bad += not ok
            for o in (s0, s1, w0):
                o.close()
        except Exception as e:  # noqa: BLE001
            log(f"ERR {tag}: {e}")
            bad += 1
    B.option_set("L2Z_SCHEME_B", 0)
    return bad


if __name__ == "__main__":
    print("bad:", run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0,
                      wide=len(sys.argv) > 3 and sys.argv[3] == "wide"))
