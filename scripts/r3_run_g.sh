#!/bin/bash
# round 3, GPU call G: is the decode path where round 2 left it?  Round-2 library (built from 86779ae) against the
# current one on the SAME box, alternating processes: per-kind kernel durations and the greedy rate.
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
{
for i in 1 2; do
  echo "--- round-2 library"; L2Z_LIB=$PWD/scripts/xlib/libllama2_hip_r02.so python scripts/kind_scan.py llama2-7b ""
  echo "--- current library"; python scripts/kind_scan.py llama2-7b ""
done
echo "--- round-2 library"; L2Z_LIB=$PWD/scripts/xlib/libllama2_hip_r02.so python scripts/ab.py llama2-7b 255 3 ""
echo "--- current library"; python scripts/ab.py llama2-7b 255 3 ""
echo "--- round-2 library"; L2Z_LIB=$PWD/scripts/xlib/libllama2_hip_r02.so python scripts/ab.py stories15M 255 3 ""
echo "--- current library"; python scripts/ab.py stories15M 255 3 ""
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -v "^$" | head -30
} > $O/r03_lib_ab.txt 2>&1
cat $O/r03_lib_ab.txt
