#!/bin/bash
# round 3, GPU call B: head-major KV cache + contiguous split chunks + attention form by position:
# the whole gpu-marked suite, the bench line, interleaved A/B of the short-context form, attention scan
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rA --durations=8 > gpurun_out/r03b_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03b_pytest_gpu.log
grep -E "passed|failed" gpurun_out/r03b_pytest_gpu.log | tail -n 3
timeout 400 python bench.py > gpurun_out/r03b_bench.json 2> gpurun_out/r03b_bench.err
{
python scripts/ab.py stories110M 255 5 "" "L2Z_ATTN_SHORT_POS=0" "L2Z_ATTN_SHORT_POS=128"
python scripts/ab.py llama2-7b 255 3 "" "L2Z_ATTN_SHORT_POS=0" "L2Z_ATTN_SHORT_POS=64" "L2Z_ATTN_SHORT_POS=256"
python scripts/attn_time_scan.py llama2-7b 0 63 127 255 256 511 1023 2047
python scripts/attn_time_scan.py stories110M 0 63 127 255 256 1023
} > gpurun_out/r03b_ab.txt 2>&1
tail -n 30 gpurun_out/r03b_ab.txt
