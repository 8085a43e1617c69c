#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_gpu_perf_gate.py -q -s 2>&1 | grep -E "common factor|passed|failed|attention at|solo rank" | cut -c1-300
for n in 64 128 512 1024; do python scripts/prefill_ab.py llama2-7b $n 5 ""; done
rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | head -5
} > gpurun_out/r6_57_gate_on_another_box.txt 2>&1
cat gpurun_out/r6_57_gate_on_another_box.txt
