#!/bin/bash
# round 5 final measurements: what profiles/r05_final_* and DESIGN.md quote.  Measurements first (a bench run straight
# after the test suite reads ~3 % low: the chip's state), the gpu-marked suite last.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/r05_final_bench.json 2> $O/r05_final_bench.err; tail -c 300 $O/r05_final_bench.err
python bench.py --steps 20 --warmup 5 > $O/r05_final_bench_driver_args.json 2>> $O/r05_final_bench.err
python - <<'PY'
import json
for f in ("r05_final_bench.json", "r05_final_bench_driver_args.json"):
    o = json.load(open("gpurun_out/" + f)); r = o["roofline"]; e = o.get("extra", {})
    print(f, "value", round(o["value"], 2), "frac", round(r["frac"], 4), "whole", round(r["whole_token_frac"], 4), "repeats", (e.get("repeats") or {}).get("median"))
    print("  b2b", {k: round(v["ms_per_launch"] * 1e3, 2) for k, v in r["by_kind_back_to_back"].items()})
    print("  prefill", (e.get("prefill") or {}).get("ms"), (e.get("prefill") or {}).get("ms_by_prompt_tokens"), "long", (e.get("long_context") or {}).get("tokens_per_s"),
          "110M", (e.get("stories110M") or {}).get("tokens_per_s"), "15M", e.get("stories15M_tokens_per_s"), "cpu", (o.get("cpu_baseline") or {}).get("value"))
    print("  solo", json.dumps((e.get("scaling_model") or {}).get("solo_rank"))[:600])
PY
# rocprofv3 kernel table of the 7B decode (the kernel the roofline object is about) and of the small shape
( cd /tmp
for wl in stories15M llama2-7b; do
  rm -rf /tmp/prof_$wl
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$wl -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 64 --warmup 2 --no-cpu-baseline --no-extra > /tmp/prof_$wl.log 2>&1 || tail -5 /tmp/prof_$wl.log
  python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/prof_$wl -name "*.db" | head -1) "round 5 (r05): rocprofv3 --kernel-trace --stats -- python bench.py --workload $wl --steps 64 --warmup 2 --no-cpu-baseline --no-extra" > $GRAFT_REPO_ROOT/$O/r05_final_${wl}_kernel_stats.md
done
for n in 48 512; do
  rm -rf /tmp/prof_pf$n
  rocprofv3 --kernel-trace --stats -d /tmp/prof_pf$n -o p -- python $GRAFT_REPO_ROOT/scripts/prefill_prof.py llama2-7b $n > /tmp/prof_pf$n.log 2>&1 || tail -5 /tmp/prof_pf$n.log
  python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/prof_pf$n -name "*.db" | head -1) "round 5 (r05): rocprofv3 --kernel-trace --stats -- python scripts/prefill_prof.py llama2-7b $n (3 prefills)" > $GRAFT_REPO_ROOT/$O/r05_final_prefill${n}_llama2-7b.md
done )
head -14 $O/r05_final_llama2-7b_kernel_stats.md
bash scripts/pmc_traffic.sh r05 > $O/r05_final_pmc.log 2>&1; tail -9 $O/r05_final_pmc.log
( for n in 40 48; do timeout 300 python scripts/prefill_ab.py llama2-7b $n 5 "" "L2Z_PF_PANEL=0"; done ) > $O/r05_final_prefill_ab.txt 2>&1; cat $O/r05_final_prefill_ab.txt
timeout 600 python -u scripts/solo_rank.py llama2-7b 128 > $O/r05_solo_rank.md 2>&1; cat $O/r05_solo_rank.md
# four ranks on this one GPU, all five legs (a proxy for the control path and for the structures' ranking)
L2Z_BENCH_LEG_TIMEOUT_S=240 timeout 1100 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 \
  --master-port 29533 bench.py --gpus 4 --steps 64 --warmup 1 > $O/r05_bench_4ranks_1gpu.json 2> $O/r05_bench_4ranks_1gpu.err
echo "4-rank bench rc=$?"
python - <<'PY'
import json
try:
    o = json.load(open("gpurun_out/r05_bench_4ranks_1gpu.json"))
    print("4 ranks:", o.get("value"), [(l["transport"], l["ok"], round(l.get("tokens_per_s") or 0, 1), (l.get("prefill_sharded") or {}).get("ms"), l.get("why")) for l in o["comm"]["legs"]])
except Exception as e:
    print("4-rank bench line:", e)
PY
timeout 2400 python -m pytest tests -m gpu -q -rA --durations=8 > $O/r05_final_pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/r05_final_pytest_gpu.log
grep -E "passed|failed|^FAILED|^ERROR" $O/r05_final_pytest_gpu.log | tail -n 8
grep -E "max \|diff\||max \|logit|identical|margin|vs oracle|vs the stepped|host replay|scheme B|common factor|panel kernel" $O/r05_final_pytest_gpu.log | head -120 > $O/r05_final_parity_numbers.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
