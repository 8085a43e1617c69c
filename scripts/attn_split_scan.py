"""Split decode attention (7B shape) per layer, back to back (l2z_time_kind), by chunk count and block size at a few
positions: is the fixed cost of the split form a function of how many blocks share a head?
usage: attn_split_scan.py [workload] [pos ...]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
args = sys.argv[1:]
wl = args.pop(0) if args and not args[0].isdigit() else "llama2-7b"
cfg, shared = {n: (c, sh) for n, c, sh in ck.iter_configs()}[wl]
w = B.Weights(cfg, None, shared, seed=1)
poss = [int(p) for p in args if int(p) < cfg.seq_len] or [256, 511, 1023, 2047]
print(f"{wl} split attention per layer (us), back to back; rows: chunks per head x threads per block")
print("            " + "".join(f"pos {p:5d} " for p in poss))
for nch in (2, 4, 8, 16):
    for nt in (256, 1024):
        B.option_set("L2Z_ATTN_SPLIT", nch)
        B.option_set("L2Z_ATTN_BLOCK", nt)
        s = B.RunState(cfg)
        row = []
        for p in poss:
            row.append(np.median([s.time_kind("attn", p, w, reps=4)[0] * 1e3 for _ in range(3)]))
        s.close()
        print(f"{nch:2d} x {nt:4d}   " + "".join(f"{v:9.2f} " for v in row))
B.option_set("L2Z_ATTN_BLOCK", 0)
B.option_set("L2Z_ATTN_SPLIT", 0)
for name, short in (("one block per head, 1024 threads", 0), ("one block per head, 256 threads (speculative first round)", cfg.seq_len)):
    B.option_set("L2Z_ATTN_SHORT_POS", short)
    s2 = B.RunState(cfg)
    print(f"{name}: " + "".join(f"{np.median([s2.time_kind('attn', p, w, reps=4)[0] * 1e3 for _ in range(3)]):9.2f} " for p in poss))
    s2.close()
B.option_set("L2Z_ATTN_SPLIT", -1); B.option_set("L2Z_ATTN_SHORT_POS", -1)
