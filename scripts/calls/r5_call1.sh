#!/bin/bash
# round 5, first GPU call: the tree with round 4's opt-in forms removed -- headline with the driver's arguments, rocprofv3
# kernel table of the 7B decode (wo / ffn2 back at their round-3 durations?), then the gpu-marked suite.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > $O/r05a_bench_driver_args.json 2> $O/r05a_bench.err; tail -c 300 $O/r05a_bench.err
python - <<'PY'
import json
o = json.load(open("gpurun_out/r05a_bench_driver_args.json"))
r = o["roofline"]
print("value", o["value"], "ms", o["ms_per_step"], "frac", r["frac"], "whole", r["whole_token_frac"])
print({k: round(v["ms_per_launch"] * 1e3, 2) for k, v in r["by_kind_back_to_back"].items()})
e = o.get("extra", {})
print("prefill", e.get("prefill", {}).get("ms"), e.get("prefill", {}).get("ms_by_prompt_tokens"), "long", (e.get("long_context") or {}).get("tokens_per_s"),
      "110M", e.get("stories110M", {}).get("tokens_per_s"), "15M", e.get("stories15M_tokens_per_s"))
print("solo", json.dumps((e.get("scaling_model") or {}).get("solo_rank")))
PY
( cd /tmp
  rm -rf /tmp/prof_7b
  rocprofv3 --kernel-trace --stats -d /tmp/prof_7b -o p -- python $GRAFT_REPO_ROOT/bench.py --workload llama2-7b --steps 64 --warmup 2 --no-cpu-baseline --no-extra > /tmp/prof_7b.log 2>&1 || tail -5 /tmp/prof_7b.log
  python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/prof_7b -name "*.db" | head -1) "round 5 (r05a): rocprofv3 --kernel-trace --stats -- python bench.py --workload llama2-7b --steps 64 --warmup 2 --no-cpu-baseline --no-extra" > $GRAFT_REPO_ROOT/$O/r05a_llama2-7b_kernel_stats.md )
head -14 $O/r05a_llama2-7b_kernel_stats.md
timeout 1500 python -m pytest tests -m gpu -q -x -rA --durations=8 > $O/r05a_pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/r05a_pytest_gpu.log
grep -E "passed|failed|^FAILED|^ERROR|us per launch" $O/r05a_pytest_gpu.log | tail -n 12
