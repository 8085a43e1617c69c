#!/bin/bash
# round 3, GPU call A: the whole gpu-marked suite (new oracle-pinned prefill tests, CLI replay tests, bench legs
# on one GPU), then the N=1 bench line and the 2- and 4-rank legs on the one GPU
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rA --durations=20 > gpurun_out/r03a_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03a_pytest_gpu.log
timeout 400 python bench.py > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err
for n in 2 4; do
  timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2955$n \
      bench.py --gpus $n --steps 128 --warmup 1 > gpurun_out/r03a_mp$n.json 2> gpurun_out/r03a_mp$n.err
done
tail -n 5 gpurun_out/r03a_pytest_gpu.log
