#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_perf_gate.py -q 2>&1 | tail -2; done > gpurun_out/r6_69_gates_x3.txt 2>&1
cat gpurun_out/r6_69_gates_x3.txt
