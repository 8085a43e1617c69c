#!/bin/bash
# round 3 final measurements: everything profiles/r03_* and DESIGN.md quote.  Measurements first (a bench run
# straight after the test suite reads ~3 % low: the chip's state), the gpu-marked suite last.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/r03_bench.json 2> $O/r03_bench.err; tail -c 300 $O/r03_bench.err
python bench.py --steps 20 --warmup 5 > $O/r03_bench_driver_args.json 2>> $O/r03_bench.err
{
python scripts/kind_scan.py llama2-7b "" "L2Z_ROW_TAIL_SKIP=0"
python scripts/ab.py llama2-7b 255 3 "" "L2Z_ROW_TAIL_SKIP=0"
} > $O/r03_tail_skip_ab.txt 2>&1; cat $O/r03_tail_skip_ab.txt
# rocprofv3 kernel tables: decode of the three shapes, prefill at 16 / 64 / 128 / 512 tokens
( cd /tmp
for wl in stories15M stories110M llama2-7b; do
  rm -rf /tmp/prof_$wl
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$wl -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 64 --warmup 2 --no-cpu-baseline --no-extra > /tmp/prof_$wl.log 2>&1 || tail -5 /tmp/prof_$wl.log
  python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/prof_$wl -name "*.db" | head -1) "round 3 (r03): rocprofv3 --kernel-trace --stats -- python bench.py --workload $wl --steps 64 --warmup 2 --no-cpu-baseline --no-extra" > $GRAFT_REPO_ROOT/$O/r03_${wl}_kernel_stats.md
done
for n in 16 64 128 512; do
  rm -rf /tmp/prof_pf
  rocprofv3 --kernel-trace --stats -d /tmp/prof_pf -o p -- python $GRAFT_REPO_ROOT/scripts/prefill_prof.py llama2-7b $n > /tmp/pf.log 2>&1 || tail -3 /tmp/pf.log
  python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/prof_pf -name "*.db" | head -1) "round 3 (r03): rocprofv3 --kernel-trace --stats -- python scripts/prefill_prof.py llama2-7b $n (3 prefills)" > $GRAFT_REPO_ROOT/$O/r03_prefill${n}_llama2-7b.md
done )
head -12 $O/r03_llama2-7b_kernel_stats.md
bash scripts/pmc_traffic.sh r03 > $O/r03_pmc.log 2>&1; tail -9 $O/r03_pmc.log
timeout 1500 python -m pytest tests -m gpu -q -rA --durations=8 > $O/r03_pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/r03_pytest_gpu.log
grep -E "passed|failed|^FAILED" $O/r03_pytest_gpu.log | tail -n 8
grep -E "max \|diff\||max \|logit|identical|margin|vs oracle|vs the stepped|host replay" $O/r03_pytest_gpu.log | head -80 > $O/r03_parity_numbers.txt
