#!/bin/bash
# round 3, GPU call H: per-launch split-K policy (65-128 tokens: block-starved launches only) -- parity subset + A/B
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
{
for n in 100 128; do python scripts/prefill_ab.py llama2-7b $n 4 "" "L2Z_PF_SPLITK=1" "L2Z_PF_SPLITK=2"; done
python scripts/prefill_ab.py llama2-7b 64 4 "" "L2Z_PF_SPLITK=1"
} > $O/r03h_ab.txt 2>&1
cat $O/r03h_ab.txt
timeout 900 python -m pytest tests -m gpu -q -rA -k "prefill or fuzz or sharded or bench_line" --deselect tests/test_gpu_fullsize.py::test_stories110M_prefill_paths_vs_oracle > $O/r03h_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/r03h_pytest_gpu.log
grep -E "passed|failed|^FAILED" $O/r03h_pytest_gpu.log | tail -n 8
( cd /tmp; rm -rf /tmp/prof_pf; rocprofv3 --kernel-trace --stats -d /tmp/prof_pf -o p -- python $GRAFT_REPO_ROOT/scripts/prefill_prof.py llama2-7b 128 > /tmp/pf.log 2>&1 || tail -3 /tmp/pf.log
  python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/prof_pf -name "*.db" | head -1) "round 3 (r03, final policy): rocprofv3 --kernel-trace --stats -- python scripts/prefill_prof.py llama2-7b 128 (3 prefills)" > $GRAFT_REPO_ROOT/$O/r03_prefill128_llama2-7b.md )
head -8 $O/r03_prefill128_llama2-7b.md | cut -c1-140
