#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for n in 64 100 128 512; do python scripts/prefill_ab.py llama2-7b $n 5 "L2Z_PF_FUSE_PLANES=0" ""; done
for e in 3 4; do
  echo "ring of $e buffers at two token tiles:"; L2Z_LIB=$PWD/llama2.zig_amd/exp/libl2z_nb$e.so python scripts/prefill_ab.py llama2-7b 64 3 ""
done
timeout 1200 python -m pytest tests/test_gpu_x3.py -q -x 2>&1 | tail -5
timeout 1200 python -m pytest tests -m gpu -q -x -k "prefill and not perf" 2>&1 | tail -3
} > gpurun_out/r6_34_fused_planes.txt 2>&1
cat gpurun_out/r6_34_fused_planes.txt
