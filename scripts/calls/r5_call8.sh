#!/bin/bash
# round 5, eighth GPU call: the final tree once more -- whole gpu suite, headline with the driver's arguments, the
# two-rank bench line on one GPU (stdout must be ONE line)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > $O/r05_final2_bench_driver_args.json 2> $O/r05_final2_bench.err
python -c "
import json; o=json.load(open('gpurun_out/r05_final2_bench_driver_args.json')); r=o['roofline']; e=o['extra']
print('value', round(o['value'],2), 'frac', round(r['frac'],4), 'whole', round(r['whole_token_frac'],4), 'prefill', e['prefill']['ms_by_prompt_tokens'], 'long', e['long_context']['tokens_per_s'])"
L2Z_BENCH_LEG_TIMEOUT_S=200 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 32 --warmup 1 --workload stories110M > $O/r05_bench_2ranks_110M.json 2> $O/r05_bench_2ranks_110M.err
echo "2-rank bench rc=$? lines on stdout: $(wc -l < $O/r05_bench_2ranks_110M.json)"
python -c "
import json; o=json.loads(open('gpurun_out/r05_bench_2ranks_110M.json').read()); print('2 ranks 110M:', o['value'], [(l['transport'], l['ok'], round(l.get('tokens_per_s') or 0), (l.get('prefill_sharded') or {}).get('ms')) for l in o['comm']['legs']])"
timeout 2400 python -m pytest tests -m gpu -q -rA --durations=8 > $O/r05_final2_pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/r05_final2_pytest_gpu.log
grep -E "passed|failed|^FAILED|^ERROR|panel prefill vs HF" $O/r05_final2_pytest_gpu.log | tail -n 8
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
