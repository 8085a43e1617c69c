#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
bash scripts/pf_prof.sh llama2-7b 64 > gpurun_out/r6_32_prefill64_kernels.md 2>&1; head -12 gpurun_out/r6_32_prefill64_kernels.md
bash scripts/pf_prof.sh llama2-7b 128 > gpurun_out/r6_32_prefill128_kernels.md 2>&1; head -12 gpurun_out/r6_32_prefill128_kernels.md
timeout 1500 python -m pytest tests -m gpu -q -x -k "prefill" 2>&1 | tail -5 > gpurun_out/r6_32_pytest_prefill.txt; cat gpurun_out/r6_32_pytest_prefill.txt
