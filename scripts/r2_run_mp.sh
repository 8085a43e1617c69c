#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
for n in 2 4 8; do for t in p2p-consume p2p-gather; do
  echo "== gpus $n transport $t"
  L2Z_COMM=$t timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2959$n bench.py --gpus $n --steps 100 --no-cpu-baseline 2>$O/r2mp_${n}_$t.err | tail -1 > $O/r2mp_${n}_$t.json
  python -c "import sys,json; d=json.loads(open('$O/r2mp_${n}_$t.json').read()); print(round(d['value'],1), 'tok/s', round(d['ms_per_step'],3), 'ms/step', d['comm']['gather_launches_per_token'], {k:round(v['ms_per_launch']*1e3,1) for k,v in d['roofline']['by_kind'].items()})" || tail -5 $O/r2mp_${n}_$t.err
done; done 2>&1 | tee $O/r2mp.txt
