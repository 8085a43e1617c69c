#!/bin/bash
# round 5, seventh GPU call: the attention of short contexts in the tail of the q | k | v launch -- correctness, then A/B
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -rA -k "attention_in_the_qkv or transformer_logits or greedy_token or 7b_device_loop or 7b_full_forward or sharded_hip_path or multiprocess_peer_write or solo_rank" > $O/r05g_pytest_new.log 2>&1; echo "pytest(new) rc=$?" | tee -a $O/r05g_pytest_new.log
grep -E "passed|failed|^FAILED|^ERROR|Error" $O/r05g_pytest_new.log | tail -n 12
timeout 400 python scripts/ab.py llama2-7b 100 4 "" "L2Z_FOLD_ATTN=0" > $O/r05g_fold_ab.txt 2>&1; cat $O/r05g_fold_ab.txt
timeout 300 python - <<'PY' 2>&1 | tail -6
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
cfg, shared = {n: (c, sh) for n, c, sh in ck.iter_configs()}["llama2-7b"]
w = B.Weights(cfg, None, shared, seed=2024)
for fold in (1, 0, 1, 0):
    B.option_set("L2Z_FOLD_ATTN", fold)
    s = B.RunState(cfg)
    B.option_set("L2Z_FOLD_ATTN", 1)
    for pos in (8, 100):
        q = min(s.time_kind("qkv", pos, w, reps=4)[0] for _ in range(3)) * 1e3
        a = min(s.time_kind("attn", pos, w, reps=4)[0] for _ in range(3)) * 1e3
        print(f"fold {fold} pos {pos}: qkv {q:.2f} us + attn {a:.2f} us = {q + a:.2f}")
    s.close()
PY
for k in 1 2; do
  echo "== solo N=8 form $k folded"; timeout 200 python -u scripts/solo_rank.py llama2-7b 128 8 $k 2>&1 | tail -2
  echo "== solo N=8 form $k, chain"; L2Z_FOLD_ATTN=0 timeout 200 python -u scripts/solo_rank.py llama2-7b 128 8 $k 2>&1 | tail -2
done > $O/r05g_solo_fold.txt 2>&1; cat $O/r05g_solo_fold.txt
python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read()); r=o['roofline']; print('driver args:', round(o['value'],2), {k: round(v['ms_per_launch']*1e3,2) for k,v in r['by_kind_back_to_back'].items()})"
