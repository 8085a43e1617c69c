#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
bash scripts/pf_prof.sh llama2-7b 64 > gpurun_out/r6_30_prefill64_kernels.md 2>&1; head -9 gpurun_out/r6_30_prefill64_kernels.md
bash scripts/pf_prof.sh llama2-7b 128 > gpurun_out/r6_30_prefill128_kernels.md 2>&1; head -9 gpurun_out/r6_30_prefill128_kernels.md
