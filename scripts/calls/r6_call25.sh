#!/bin/bash
# round 6, call 25: the perf floors again after the bf16-core prefill kernels (two runs)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
python scripts/perf_floor.py > $O/r6_25_perf_floor_a.json 2> $O/r6_25_perf_floor_a.err
python scripts/perf_floor.py > $O/r6_25_perf_floor_b.json 2> $O/r6_25_perf_floor_b.err
python - <<'PY'
import json
a=json.load(open("gpurun_out/r6_25_perf_floor_a.json")); b=json.load(open("gpurun_out/r6_25_perf_floor_b.json"))
print("prefill a", a["prefill_ms"]); print("prefill b", b["prefill_ms"])
print("7b decode a", a["decode_us_per_launch"]["llama2-7b"]); print("7b decode b", b["decode_us_per_launch"]["llama2-7b"])
print("attn", a["attention_us_per_layer_pos2047"], b["attention_us_per_layer_pos2047"], "solo", a["solo_rank_tokens_per_s"], b["solo_rank_tokens_per_s"])
print("tok/s", a["decode_tokens_per_s"], b["decode_tokens_per_s"])
PY
