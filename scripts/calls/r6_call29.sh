#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
timeout 300 python scripts/x3_accuracy.py llama2-7b 128
for n in 128 100 64 49; do
timeout 300 python scripts/prefill_ab.py llama2-7b $n 5 ""
done
} > gpurun_out/r6_29_stream_coop.txt 2>&1
cat gpurun_out/r6_29_stream_coop.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "prefill or x3 or bf16" --deselect tests/test_gpu_perf_gate.py 2>&1 | tail -5
