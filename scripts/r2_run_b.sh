#!/bin/bash
# round 2, GPU call B
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --timeout 900 > $O/r2b_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/r2b_pytest.log
tail -30 $O/r2b_pytest.log
grep -E "max \|diff\||max \|logit|identical|margin" $O/r2b_pytest.log | head -40
./scripts/ll_poll_probe 2>&1 | tee $O/r2b_ll_poll.txt
python scripts/ab.py stories15M 255 5 "" "L2Z_FUSE_SMALL=0" "L2Z_NT_SMALL=1" "L2Z_FUSE_SMALL=0,L2Z_NT_SMALL=1" 2>&1 | tee $O/r2b_ab15.txt
python scripts/ab.py stories110M 255 4 "" "L2Z_NT_SMALL=0" 2>&1 | tee $O/r2b_ab110.txt
python scripts/ab.py llama2-7b 64 3 1900 "" "L2Z_ATTN_BLOCK=256" "L2Z_ATTN_SPLIT=16" 2>&1 | tee $O/r2b_ab7b_long.txt
python scripts/ab.py llama2-7b 128 3 "" "L2Z_ROW_BLOCKS=1" 2>&1 | tee $O/r2b_ab7b.txt
python scripts/attn_scan.py 2>&1 | tee $O/r2b_attn_scan.txt
for n in 2 4; do for t in p2p-consume p2p-gather; do
  echo "== gpus $n transport $t"
  L2Z_COMM=$t timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2958$n bench.py --gpus $n --steps 100 --no-cpu-baseline 2>$O/r2b_mp_${n}_$t.err | tail -1 > $O/r2b_mp_${n}_$t.json
  python -c "import sys,json; d=json.loads(open('$O/r2b_mp_${n}_$t.json').read()); print(d['value'], d['ms_per_step'], {k:round(v['ms_per_launch']*1e3,1) for k,v in d['roofline']['by_kind'].items()})" || tail -5 $O/r2b_mp_${n}_$t.err
done; done 2>&1 | tee $O/r2b_mp.txt
