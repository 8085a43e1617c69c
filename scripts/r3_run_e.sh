#!/bin/bash
# round 3, GPU call E: the whole gpu suite on the final split-K policy + bench line
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rA --durations=8 > gpurun_out/r03e_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03e_pytest_gpu.log
grep -E "passed|failed|^FAILED" gpurun_out/r03e_pytest_gpu.log | tail -n 12
timeout 400 python bench.py > gpurun_out/r03e_bench.json 2> gpurun_out/r03e_bench.err
python scripts/prefill_ab.py llama2-7b 60 4 "" "L2Z_PF_SPLITK=1" > gpurun_out/r03e_ab.txt 2>&1
python scripts/prefill_ab.py llama2-7b 100 4 "" "L2Z_PF_SPLITK=1" >> gpurun_out/r03e_ab.txt 2>&1
cat gpurun_out/r03e_ab.txt
