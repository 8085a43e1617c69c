#!/bin/bash
# round 2 final measurements: everything profiles/ and DESIGN.md quote (tag r02)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --timeout 900 > $O/r02_pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/r02_pytest_gpu.log
tail -4 $O/r02_pytest_gpu.log
grep -E "max \|diff\||max \|logit|identical|margin|attention .* max" $O/r02_pytest_gpu.log | head -40 > $O/r02_parity_numbers.txt
python bench.py > $O/r02_bench.json 2> $O/r02_bench.err; tail -c 400 $O/r02_bench.err
python bench.py --steps 20 --warmup 5 > $O/r02_bench_driver_args.json 2>> $O/r02_bench.err
bash scripts/r2_prof.sh r02 > $O/r02_prof.log 2>&1
bash scripts/pmc_traffic.sh r02 > $O/r02_pmc.log 2>&1; tail -12 $O/r02_pmc.log
( cd /tmp; rm -rf /tmp/prof_pf; rocprofv3 --kernel-trace --stats -d /tmp/prof_pf -o p -- python $GRAFT_REPO_ROOT/scripts/prefill_prof.py llama2-7b 512 > /tmp/pf.log 2>&1; python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/prof_pf -name "*.db" | head -1) "round 2 (r02): rocprofv3 --kernel-trace --stats -- python scripts/prefill_prof.py llama2-7b 512 (3 prefills)" > $GRAFT_REPO_ROOT/$O/r02_prefill512_llama2-7b.md )
python scripts/ab.py stories15M 255 5 "" "L2Z_FUSE_SMALL=0" 2>&1 | tee $O/r02_ab.txt
python scripts/ab.py stories110M 255 4 "" 2>&1 | tee -a $O/r02_ab.txt
python scripts/ab.py llama2-7b 128 3 "" "L2Z_ROW_BLOCKS=1" "L2Z_ROW_BLOCKS=4" "L2Z_ATTN_BLOCK=256" 2>&1 | tee -a $O/r02_ab.txt
python scripts/ab.py llama2-7b 64 3 1900 "" "L2Z_ATTN_BLOCK=256" "L2Z_ATTN_SPLIT=16" 2>&1 | tee -a $O/r02_ab.txt
python scripts/prefill_ab.py llama2-7b 512 4 "" "L2Z_PF_FUSE=0" "L2Z_PF_DMA=0" 2>&1 | tee $O/r02_prefill_ab.txt
python scripts/prefill_ab.py llama2-7b 256 4 "" "L2Z_PF_DMA=0" 2>&1 | tee -a $O/r02_prefill_ab.txt
python scripts/prefill_ab.py llama2-7b 64 4 "" 2>&1 | tee -a $O/r02_prefill_ab.txt
python scripts/attn_scan.py 2>&1 | tee $O/r02_attn_scan.txt
python scripts/kind_scan.py llama2-7b "" "L2Z_ROW_BLOCKS=1" "L2Z_ROW_BLOCKS=4" 2>&1 | tee $O/r02_kind_scan.txt
./scripts/ll_poll_probe 2>&1 | tee $O/r02_ll_poll_probe.txt
./scripts/xcd_affinity_probe 2>&1 | tee $O/r02_xcd_affinity_probe.txt
python scripts/upload_rate.py 2>&1 | tee $O/r02_upload_rate.txt
for n in 2 4; do for t in p2p-consume p2p-gather; do
  L2Z_COMM=$t timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2960$n bench.py --gpus $n --steps 100 --no-cpu-baseline 2>$O/r02_mp_${n}_$t.err | tail -1 > $O/r02_mp_${n}_$t.json
  python -c "import sys,json; d=json.loads(open('$O/r02_mp_${n}_$t.json').read()); print('gpus $n (all ranks on ONE GPU) $t:', round(d['value'],1), 'tok/s', round(d['ms_per_step'],3), 'ms/step', {k:round(v['ms_per_launch']*1e3,1) for k,v in d['roofline']['by_kind'].items()})" || tail -3 $O/r02_mp_${n}_$t.err
done; done 2>&1 | tee $O/r02_mp.txt
