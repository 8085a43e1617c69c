#!/bin/bash
# round 3, GPU call D: split-K family of the tile GEMM -- parity (prefill, sharded, fuzz), interleaved A/B by prompt length
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -rA -k "prefill or fuzz or golden or sharded" --deselect tests/test_gpu_fullsize.py::test_stories110M_prefill_paths_vs_oracle > gpurun_out/r03d_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03d_pytest_gpu.log
grep -E "passed|failed|^FAILED" gpurun_out/r03d_pytest_gpu.log | tail -n 12
{
for n in 100 128 200 256; do python scripts/prefill_ab.py llama2-7b $n 4 "L2Z_PF_SPLITK=1" "" "L2Z_PF_SPLITK=2" "L2Z_PF_SPLITK=4"; done
for n in 40 64; do python scripts/prefill_ab.py llama2-7b $n 4 "" "L2Z_PF_SKINNY_MAX=32,L2Z_PF_SPLITK=1" "L2Z_PF_SKINNY_MAX=32,L2Z_PF_SPLITK=2" "L2Z_PF_SKINNY_MAX=32,L2Z_PF_SPLITK=4"; done
python scripts/prefill_ab.py stories110M 128 4 "L2Z_PF_SPLITK=1" "" "L2Z_PF_SPLITK=2" "L2Z_PF_SPLITK=4"
python scripts/prefill_ab.py stories110M 64 4 "" "L2Z_PF_SKINNY_MAX=32,L2Z_PF_SPLITK=1" "L2Z_PF_SKINNY_MAX=32,L2Z_PF_SPLITK=2" "L2Z_PF_SKINNY_MAX=32,L2Z_PF_SPLITK=4"
} > gpurun_out/r03d_ab.txt 2>&1
cat gpurun_out/r03d_ab.txt
