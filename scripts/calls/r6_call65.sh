#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for n in 660 700 800 900; do python scripts/prefill_ab.py llama2-7b $n 3 ""; done
timeout 1500 python -m pytest tests -m gpu -q -x -k "prefill and not perf" 2>&1 | tail -3
timeout 900 python scripts/fuzz_prefill.py 40 81 2>&1 | grep -E "^bad:|BAD|ERR" | tail -3
python - <<'PY'
# a 7B prompt of 900 tokens (one chunk of 900 since this change) against the same prompt in chunks of 512 + 388: logits within the tolerance
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
cfg = ck.LLAMA2_7B
w = B.Weights(cfg, None, False, seed=3); s = B.RunState(cfg)
toks = [1] + np.random.default_rng(5).integers(2, cfg.vocab_size, 899).tolist()
s.prefill(toks, 0, w); a = s.logits().copy()
B.option_set("L2Z_PF_CHUNK", 512); s.prefill(toks, 0, w); b = s.logits().copy(); B.option_set("L2Z_PF_CHUNK", 0)
print("900 tokens, one chunk vs 512 + 388: max |logit diff|", float(np.abs(a - b).max()), "argmax equal", int(a.argmax()) == int(b.argmax()))
PY
} > gpurun_out/r6_65_chunk_plan_whole.txt 2>&1
cat gpurun_out/r6_65_chunk_plan_whole.txt
