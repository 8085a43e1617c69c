#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
python scripts/perf_floor.py > $O/r6_49_perf_floor_a.json 2> $O/r6_49_perf_floor_a.err; tail -2 $O/r6_49_perf_floor_a.err
python scripts/perf_floor.py > $O/r6_49_perf_floor_b.json 2> $O/r6_49_perf_floor_b.err
python -c "
import json
a=json.load(open('$O/r6_49_perf_floor_a.json')); b=json.load(open('$O/r6_49_perf_floor_b.json'))
print('prefill a', a['prefill_ms']); print('prefill b', b['prefill_ms']); print('attn', a['attention_us_per_layer_pos2047'], b['attention_us_per_layer_pos2047'])
print('tok/s', a['decode_tokens_per_s'], b['decode_tokens_per_s']); print('7b', a['decode_us_per_launch']['llama2-7b']); print('solo', a['solo_rank_tokens_per_s'], b['solo_rank_tokens_per_s'])"
for n in 64 128; do bash scripts/stream_pmc_traffic.sh $n > $O/r6_49_stream_pmc_traffic_$n.md 2>&1; cat $O/r6_49_stream_pmc_traffic_$n.md | tail -6; done
