#!/bin/bash
# round 2, GPU call A: parity first, then the bench line and A/B legs for this round's changes
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --timeout 900 > $O/r2a_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/r2a_pytest.log
tail -25 $O/r2a_pytest.log
python bench.py > $O/r2a_bench.json 2> $O/r2a_bench.err; tail -c 2500 $O/r2a_bench.json
for kv in "L2Z_ATTN_PREFETCH=0" "L2Z_ATTN_PREFETCH=50" "L2Z_CLS_HANDOVER=0" "L2Z_ROW_BLOCKS=3"; do
  echo "== $kv"; env $kv python bench.py --no-extra --no-cpu-baseline --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], {k:round(v['ms_per_launch']*1e3,1) for k,v in d['roofline']['by_kind'].items()})"
done 2>&1 | tee $O/r2a_ab.txt
for wl in stories15M stories110M; do
  for kv in "X=1" "L2Z_CLS_HANDOVER=0"; do
  echo "== $wl $kv"; env $kv python bench.py --workload $wl --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], {k:round(v['ms_per_launch']*1e3,1) for k,v in d['roofline']['by_kind'].items()})"
  done
done 2>&1 | tee $O/r2a_small.txt
python scripts/attn_scan.py 2>&1 | tee $O/r2a_attn_scan.txt
for n in 2 4; do for t in p2p-consume p2p-gather; do
  echo "== gpus $n transport $t"
  L2Z_COMM=$t timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2957$n bench.py --gpus $n --steps 100 --no-cpu-baseline 2>$O/r2a_mp_${n}_$t.err | tail -1 > $O/r2a_mp_${n}_$t.json
  python -c "import sys,json; d=json.loads(open('$O/r2a_mp_${n}_$t.json').read()); print(d['value'], d['ms_per_step'], d['comm'], {k:round(v['ms_per_launch']*1e3,1) for k,v in d['roofline']['by_kind'].items()})" || tail -5 $O/r2a_mp_${n}_$t.err
done; done 2>&1 | tee $O/r2a_mp.txt
