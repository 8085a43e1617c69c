#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for n in 64 100 128 256 512 1024; do python scripts/prefill_ab.py llama2-7b $n 5 ""; done
L2Z_LIB=$PWD/llama2.zig_amd/exp/libl2z_x3tl.so timeout 300 python scripts/x3_timeline.py llama2-7b 128 | grep -E "^[qW]|epilogue"
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -x -k "7b_prefill" -s 2>&1 | grep -E "7B prefill|passed|failed|Error" | tail -12
timeout 1200 python -m pytest tests/test_gpu_x3.py -q -x 2>&1 | tail -3
timeout 1200 python -m pytest tests -m gpu -q -x -k "prefill and not perf and not 7b" 2>&1 | tail -3
} > gpurun_out/r6_45_epilogue_hoisted.txt 2>&1
cat gpurun_out/r6_45_epilogue_hoisted.txt
