#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_gpu_perf_gate.py > gpurun_out/r6_20_pytest_gpu.txt 2>&1
tail -40 gpurun_out/r6_20_pytest_gpu.txt
