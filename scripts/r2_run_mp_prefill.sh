#!/bin/bash
# bench.py --gpus N with all ranks on ONE GPU (the box has one): the sharded-prefill leg of comm{}
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
for n in 2 4; do
  echo "== gpus $n"
  timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2969$n bench.py --gpus $n --steps 64 --no-cpu-baseline 2>$O/r2mpp_$n.err | tail -1 > $O/r2mpp_$n.json
  python -c "import sys,json; d=json.loads(open('$O/r2mpp_$n.json').read()); print(round(d['value'],1), 'tok/s', d['comm']['transport'], d['comm']['prefill_sharded'])" || tail -5 $O/r2mpp_$n.err
done 2>&1 | tee $O/r2mpp.txt
