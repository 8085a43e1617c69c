#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for n in 256 512 1024; do python scripts/prefill_ab.py llama2-7b $n 5 "L2Z_PF_FUSE_PLANES=0" ""; done
timeout 900 python -m pytest tests/test_gpu_x3.py -q -x 2>&1 | tail -3
timeout 1500 python -m pytest tests -m gpu -q -x -k "prefill and not perf" 2>&1 | tail -3
} > gpurun_out/r6_70_pair_planes.txt 2>&1
cat gpurun_out/r6_70_pair_planes.txt
