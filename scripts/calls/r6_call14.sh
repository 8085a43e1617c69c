#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
bash scripts/pf_prof.sh llama2-7b 1024 > gpurun_out/r6_14_prefill1024_x3_kernels.md 2>&1
bash scripts/pf_prof.sh llama2-7b 128 > gpurun_out/r6_14_prefill128_x3_kernels.md 2>&1
head -40 gpurun_out/r6_14_prefill1024_x3_kernels.md
head -30 gpurun_out/r6_14_prefill128_x3_kernels.md
