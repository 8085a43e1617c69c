"""Debug: one forward pass on a 2-rank shard group (two processes, one GPU) with the persistent launches against the
unsharded default chain, buffer by buffer.  usage: dbg_engine_shard.py [world]"""
import json, os, subprocess, sys, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
kw = dict(dim=4096, hidden_dim=8192, n_layers=int(os.environ.get("DBG_LAYERS", "1")), n_heads=32, n_kv_heads=8, vocab_size=8192, seq_len=320)
world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
if len(sys.argv) > 2 and sys.argv[2] == "worker":
    rank, d = int(sys.argv[3]), sys.argv[4]
    import time, __graft_entry__ as ge
    pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
    cfg = ck.Config(**kw)
    comm = B.Comm(rank, world, None, 0)
    h = comm.p2p_export(max(cfg.dim, cfg.hidden_dim, cfg.vocab_size), max(cfg.dim, cfg.hidden_dim))
    open(os.path.join(d, f"h_{rank}.tmp"), "wb").write(h); os.rename(os.path.join(d, f"h_{rank}.tmp"), os.path.join(d, f"h_{rank}.bin"))
    hs = []
    for r in range(world):
        p = os.path.join(d, f"h_{r}.bin")
        while not os.path.exists(p): time.sleep(0.01)
        hs.append(open(p, "rb").read())
    comm.p2p_connect(b"".join(hs))
    w = B.Weights(cfg, None, False, seed=33, comm=comm); s = B.RunState(cfg, comm=comm)
    s.transformer(5, 0, w)
    out = {k: s.read(k, 0, n) for k, n in (("x", cfg.dim), ("xb", cfg.dim), ("hb", cfg.hidden_dim), ("q", cfg.dim // world))}
    np.savez(os.path.join(d, f"dump_{rank}.npz"), logits=s.logits(), **out)
    sys.exit(0)
import __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
cfg = ck.Config(**kw)
d = tempfile.mkdtemp()
env = dict(os.environ, L2Z_P2P_TIMEOUT_S="5", L2Z_GRID_CAP=str(512 // world), L2Z_ENGINE=os.environ.get("DBG_ENGINE", "1"))
ps = [subprocess.Popen([sys.executable, __file__, str(world), "worker", str(r), d], env=env) for r in range(world)]
for p in ps: p.wait(timeout=120)
w = B.Weights(cfg, None, False, seed=33); s = B.RunState(cfg)
s.transformer(5, 0, w)
ref = {k: s.read(k, 0, n) for k, n in (("x", cfg.dim), ("xb", cfg.dim), ("hb", cfg.hidden_dim), ("q", cfg.dim))}
ref["logits"] = s.logits()
for r in range(world):
    o = np.load(os.path.join(d, f"dump_{r}.npz"))
    dl, hl = cfg.dim // world, cfg.hidden_dim // world
    sl = {"x": slice(r * dl, (r + 1) * dl), "xb": slice(r * dl, (r + 1) * dl), "hb": slice(r * hl, (r + 1) * hl)}
    msg = [f"rank {r}:"]
    for k in ("q", "xb", "x", "hb", "logits"):
        a = o[k][sl[k]] if k in sl else o[k]
        b = ref[k][sl[k]] if k in sl else (ref[k][r * dl:(r + 1) * dl] if k == "q" else ref[k])
        nb = int((a != b).sum())
        msg.append(f"{k} {nb} of {a.size} differ (max |d| {float(np.abs(a - b).max()):.3g})")
    print("  ".join(msg))
