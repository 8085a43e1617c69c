#!/bin/bash
# round 3, GPU call L: two-block form at long chunks (512 / 1024 / 2000 tokens), interleaved
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
{
python scripts/prefill_ab.py llama2-7b 512 3 "L2Z_PF_KGS=0" "" "L2Z_PF_KGS=10"
python scripts/prefill_ab.py llama2-7b 1024 3 "L2Z_PF_KGS=0" "" "L2Z_PF_KGS=10" "L2Z_PF_KGS=14"
python scripts/prefill_ab.py llama2-7b 2000 3 "L2Z_PF_KGS=0" ""
python scripts/prefill_ab.py llama2-7b 700 3 "L2Z_PF_KGS=0" ""
} > $O/r03l_ab.txt 2>&1
cat $O/r03l_ab.txt
