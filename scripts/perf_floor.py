"""Measure everything tests/test_gpu_perf_gate.py holds to profiles/perf_floor.json, with the SAME functions the gate
uses (it imports this file), and print the JSON:   python scripts/perf_floor.py > gpurun_out/perf_floor_new.json
Round 6: beside the 7B decode kernels (round 5) -- the launches of the small models, the batched prefill at the chunk
lengths every kernel family serves, long-context attention, and one rank of 8 alone (scheme A gather launches, scheme B)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

DECODE_SHAPES = {"llama2-7b": 8, "stories110M": 8, "stories42M": 8, "stories15M": 8}   # shape -> position timed
PREFILL_TOKENS = (16, 32, 48, 64, 96, 128, 256, 512, 1024)


def decode_kinds(B, ck, name, pos, w=None, s=None):
    """us per launch of every kind of launch of the decode pass (l2z_time_kind: the kind's launches of all layers back
    to back between one event pair), best of 3"""
    cfg, shared = {n: (c, sh) for n, c, sh in ck.iter_configs()}[name]
    own = w is None
    if own:
        w, s = B.Weights(cfg, None, shared, seed=2024), B.RunState(cfg)
    s.greedy_begin([]); s.greedy_run(w, 4)
    out = {}
    for kind in ("qkv", "attn", "wo", "ffn13", "ffn2", "cls"):
        ms, n = s.time_kind(kind, pos, w, reps=4)
        if n == 0 or ms * 1e3 < 1.0:
            continue   # (small MHA models: attention is part of the qkv launch -- nothing is launched for this kind)
        out[kind] = min([ms] + [s.time_kind(kind, pos, w, reps=4)[0] for _ in range(2)]) * 1e3
    if own:
        s.close(); w.close()
    return out


def decode_tokens_per_s(B, ck, name, steps=255):
    cfg, shared = {n: (c, sh) for n, c, sh in ck.iter_configs()}[name]
    w, s = B.Weights(cfg, None, shared, seed=2024), B.RunState(cfg)
    best = 0.0
    for _ in range(3):
        s.greedy_begin([]); s.greedy_run(w, 1); s.synchronize()
        t0 = time.perf_counter()
        n = len(s.greedy_run(w, min(steps, cfg.seq_len - 1)))
        s.synchronize()
        best = max(best, n / (time.perf_counter() - t0))
    s.close(); w.close()
    return best


def prefill_ms(B, ck, w, s, cfg, n, rounds=6):
    toks = [1] + np.random.default_rng(1).integers(2, cfg.vocab_size, n - 1).tolist()
    s.prefill(toks, 0, w)
    xs = []
    for _ in range(rounds):
        t0 = time.perf_counter(); s.prefill(toks, 0, w); xs.append(time.perf_counter() - t0)
    return min(xs) * 1e3


def attention_long_us(B, w, s, pos=2047):
    return min(s.time_kind("attn", pos, w, reps=4)[0] for _ in range(3)) * 1e3


def solo_rank(B, ck, world=8):
    import bench
    cfg = ck.LLAMA2_7B
    forms = [f for f in bench.SOLO_FORMS if f[0] in ("p2p-gather", "p2p-allreduce")]
    row = bench.solo_rank_model(B, cfg, False, 2024, 64, (world,), forms)[str(world)]
    return {k: v.get("tokens_per_s_upper_bound") for k, v in row.items()}


def measure_all(B, ck):
    name, cus, _ = B.device_info(0)
    out = {"device": {"name": name, "cus": cus}, "decode_us_per_launch": {}, "decode_tokens_per_s": {}}
    cfg = ck.LLAMA2_7B
    w, s = B.Weights(cfg, None, False, seed=2024), B.RunState(cfg)
    out["decode_us_per_launch"]["llama2-7b"] = decode_kinds(B, ck, "llama2-7b", 8, w, s)
    out["attention_us_per_layer_pos2047"] = attention_long_us(B, w, s)
    out["prefill_ms"] = {str(n): prefill_ms(B, ck, w, s, cfg, n) for n in PREFILL_TOKENS}
    s.close(); w.close()
    for nm in ("stories110M", "stories42M", "stories15M"):
        out["decode_us_per_launch"][nm] = decode_kinds(B, ck, nm, DECODE_SHAPES[nm])
        out["decode_tokens_per_s"][nm] = decode_tokens_per_s(B, ck, nm)
    out["solo_rank_tokens_per_s"] = solo_rank(B, ck, 8)
    return out


if __name__ == "__main__":
    import __graft_entry__ as ge
    pkg = ge.load_package()
    print(json.dumps(measure_all(pkg.binding, pkg.checkpoint), indent=1))
