#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x -k "x3 or prefill or bf16 or split_terms" --deselect tests/test_gpu_perf_gate.py > gpurun_out/r6_24_tests.txt 2>&1
tail -40 gpurun_out/r6_24_tests.txt
