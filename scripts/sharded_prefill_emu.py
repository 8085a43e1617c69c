"""Row-sharded prefill on emulated ranks (one GPU): wall time of l2z_emu_prefill / world ~ one rank's
launches of a sharded prefill (every stage of every rank runs back to back on the one GPU; the
device-to-device block copies and the per-stage syncs are included, the xGMI exchange is not).
   sharded_prefill_emu.py <shape> <n_tokens> [worlds...]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
shape, n = sys.argv[1], int(sys.argv[2])
worlds = [int(x) for x in sys.argv[3:]] or [2, 4, 8]
cfg, shared = {k: (c, sh) for k, c, sh in ck.iter_configs()}[shape]
toks = [1] + np.random.default_rng(1).integers(2, cfg.vocab_size, n - 1).tolist()
flops = 2.0 * n * (cfg.n_layers * (2 * cfg.dim * cfg.dim + 2 * cfg.dim * cfg.kv_dim + 3 * cfg.dim * cfg.hidden_dim))

def med(f, reps=3):
    f(); xs = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); xs.append(time.perf_counter() - t0)
    return float(np.median(xs))

w = B.Weights(cfg, None, shared, seed=1); s = B.RunState(cfg)
t1 = med(lambda: s.prefill(toks, 0, w))
ref = s.logits()
print(f"{shape} prefill {n} tokens, 1 rank: {t1*1e3:.2f} ms = {flops/t1/1e12:.1f} TFLOP/s")
s.close(); w.close()
for world in worlds:
    comms = [B.Comm(r, world, None, 0, emulated=True) for r in range(world)]
    ws = [B.Weights(cfg, None, shared, seed=1, comm=c) for c in comms]
    ss = [B.RunState(cfg, comm=c) for c in comms]
    t = med(lambda: B.emu_prefill(ss, ws, toks, 0))
    same = all(np.array_equal(x.logits(), ref) for x in ss)
    print(f"  {world} emulated ranks: {t*1e3:.2f} ms for all ranks = {t/world*1e3:.2f} ms per rank "
          f"({t1/(t/world):.2f}x of one GPU before the exchange), logits bit-identical: {same}")
    for o in ss + ws: o.close()
    for c in comms: c.close()
