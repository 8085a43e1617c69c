#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
bash scripts/pf_prof.sh llama2-7b 1024 > gpurun_out/r6_59_prefill1024_kernels.md 2>&1; head -16 gpurun_out/r6_59_prefill1024_kernels.md
bash scripts/pf_prof.sh llama2-7b 256 > gpurun_out/r6_59_prefill256_kernels.md 2>&1; head -16 gpurun_out/r6_59_prefill256_kernels.md
