#!/bin/bash
# round 6, GPU call 5: the whole GPU suite after the pruning (15 knobs, product / test libraries), perf gate included
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out; O=gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/r6_5_pytest_gpu.txt
cat $O/r6_5_pytest_gpu.txt
