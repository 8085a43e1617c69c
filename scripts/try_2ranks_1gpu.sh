#!/bin/bash
# experiment: can RCCL run 2 ranks on the same GPU?  (expected: no, "duplicate GPU")
cat > /tmp/two.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch.distributed as dist
import __graft_entry__ as ge
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
uid = [B.Comm.unique_id() if rank == 0 else None]
dist.broadcast_object_list(uid, src=0)
try:
    comm = B.Comm(rank, world, uid[0], 0)
    cfg = ck.Config(dim=64, hidden_dim=172, n_layers=2, n_heads=4, n_kv_heads=2, vocab_size=512, seq_len=32)
    w = B.Weights(cfg, None, False, seed=3, comm=comm); s = B.RunState(cfg, comm=comm)
    s.greedy_begin([]); print(rank, "tokens", s.greedy_run(w, 16))
except Exception as e:
    print(rank, "FAILED:", e)
PY
NCCL_DEBUG=WARN timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 /tmp/two.py 2>&1 | tail -15
