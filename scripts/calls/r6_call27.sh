#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q > gpurun_out/r6_27_pytest_gpu.txt 2>&1
tail -15 gpurun_out/r6_27_pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r6_27_bench.json 2> gpurun_out/r6_27_bench.err; tail -2 gpurun_out/r6_27_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6_27_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"]); print(json.dumps(d["extra"]["prefill"], indent=1)[:2500])
PY
