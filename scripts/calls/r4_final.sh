#!/bin/bash
# round 4 final measurements: what profiles/r04_final_* and DESIGN.md quote.  Measurements first (a bench run straight
# after the test suite reads ~3 % low: the chip's state), the gpu-marked suite last.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/r04_final_bench.json 2> $O/r04_final_bench.err; tail -c 300 $O/r04_final_bench.err
python bench.py --steps 20 --warmup 5 > $O/r04_final_bench_driver_args.json 2>> $O/r04_final_bench.err
# rocprofv3 kernel table of the 7B decode (the kernel the roofline object is about) and of the small shapes
( cd /tmp
for wl in stories15M llama2-7b; do
  rm -rf /tmp/prof_$wl
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$wl -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 64 --warmup 2 --no-cpu-baseline --no-extra > /tmp/prof_$wl.log 2>&1 || tail -5 /tmp/prof_$wl.log
  python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/prof_$wl -name "*.db" | head -1) "round 4 (r04): rocprofv3 --kernel-trace --stats -- python bench.py --workload $wl --steps 64 --warmup 2 --no-cpu-baseline --no-extra" > $GRAFT_REPO_ROOT/$O/r04_final_${wl}_kernel_stats.md
done )
head -14 $O/r04_final_llama2-7b_kernel_stats.md
bash scripts/pmc_traffic.sh r04 > $O/r04_final_pmc.log 2>&1; tail -9 $O/r04_final_pmc.log
# four ranks on this one GPU, all six legs (a proxy for the control path and for the structures' ranking)
L2Z_BENCH_LEG_TIMEOUT_S=240 timeout 1100 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 \
  --master-port 29533 bench.py --gpus 4 --steps 64 --warmup 1 > $O/r04_bench_4ranks_1gpu.json 2> $O/r04_bench_4ranks_1gpu.err
echo "4-rank bench rc=$?"
timeout 2400 python -m pytest tests -m gpu -q -rA --durations=8 > $O/r04_final_pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/r04_final_pytest_gpu.log
grep -E "passed|failed|^FAILED" $O/r04_final_pytest_gpu.log | tail -n 8
grep -E "max \|diff\||max \|logit|identical|margin|vs oracle|vs the stepped|host replay|scheme B" $O/r04_final_pytest_gpu.log | head -100 > $O/r04_final_parity_numbers.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
