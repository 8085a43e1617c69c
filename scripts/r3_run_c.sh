#!/bin/bash
# round 3, GPU call C: rmsnorm inside the short-prompt GEMMs -- parity (prefill tests + fuzz), interleaved A/B
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -rA -k "prefill or fuzz or rmsnorm or golden or c_abi" --deselect tests/test_gpu_fullsize.py::test_stories110M_prefill_paths_vs_oracle > gpurun_out/r03c_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03c_pytest_gpu.log
grep -E "passed|failed" gpurun_out/r03c_pytest_gpu.log | tail -n 3
{
for n in 4 16 40; do python scripts/prefill_ab.py llama2-7b $n 5 "" "L2Z_PF_RMS_FUSE=0" "L2Z_PF_ATTN=0" "L2Z_PF_RMS_FUSE=0,L2Z_PF_ATTN=0"; done
python scripts/prefill_ab.py llama2-7b 64 5 "" "L2Z_PF_ATTN=0"
python scripts/prefill_ab.py stories110M 16 5 "" "L2Z_PF_RMS_FUSE=0" "L2Z_PF_ATTN=0"
} > gpurun_out/r03c_ab.txt 2>&1
cat gpurun_out/r03c_ab.txt
