#!/bin/bash
# The GPU calls of round 4, as they were made:   gpurun -- 'bash scripts/r4_calls.sh <letter>'
# Each section writes under gpurun_out/; what mattered was copied to profiles/ (profiles/README.md says which).
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
case "$1" in
a)
# round 4, GPU call A: the overlapped decode chain (two streams, LL hand-overs, duo mat-vecs) -- does it run, are
# the bits those of the single chain, and what does it buy: interleaved A/B at the 7B shape by mode, edge set and
# hint form; then the decode parity tests
export L2Z_P2P_TIMEOUT_S=3
{
timeout 300 python scripts/ab.py llama2-7b 128 3 "" "L2Z_OVERLAP=0" "L2Z_DUO=0" "L2Z_NO_GRAPH=1" "L2Z_NO_GRAPH=1,L2Z_OVERLAP=0"
echo "rc=$?"
timeout 300 python scripts/ab.py llama2-7b 128 3 "" "L2Z_OVERLAP_EDGES=14" "L2Z_OVERLAP_EDGES=13" "L2Z_OVERLAP_EDGES=11" "L2Z_OVERLAP_EDGES=7" "L2Z_OVERLAP_EDGES=8" "L2Z_OVERLAP_HINT=0" "L2Z_OVERLAP_HINT_SLEEP=1" "L2Z_OVERLAP_HINT_SLEEP=6"
echo "rc=$?"
timeout 300 python scripts/ab.py llama2-7b 64 3 300 "" "L2Z_OVERLAP=0" "L2Z_DUO=0"
echo "rc=$?"
timeout 300 python scripts/ab.py llama2-7b 48 3 1900 "" "L2Z_OVERLAP=0" "L2Z_DUO=0"
echo "rc=$?"
} > $O/r04a_ab.txt 2>&1
cat $O/r04a_ab.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "transformer or greedy or golden or c_abi or attention or 7b" > $O/r04a_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/r04a_pytest_gpu.log
tail -n 15 $O/r04a_pytest_gpu.log
;;
esac
