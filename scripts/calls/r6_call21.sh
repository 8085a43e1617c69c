#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
python scripts/x3_accuracy.py llama2-7b 300
python scripts/x3_e2e.py 300
python scripts/x3_e2e.py 120
} > gpurun_out/r6_21_x3_e2e.txt 2>&1
cat gpurun_out/r6_21_x3_e2e.txt
