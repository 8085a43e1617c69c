#!/bin/bash
# round 6 final measurements: what profiles/r06_final_* and DESIGN.md quote.  Measurements first (a bench run straight
# after the test suite reads ~3 % low: the chip's state), fuzz sweeps and the gpu-marked suite last.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/r06_final_bench.json 2> $O/r06_final_bench.err; tail -c 300 $O/r06_final_bench.err
python bench.py --steps 20 --warmup 5 > $O/r06_final_bench_driver_args.json 2>> $O/r06_final_bench.err
python - <<'PY'
import json
for f in ("r06_final_bench.json", "r06_final_bench_driver_args.json"):
    o = json.load(open("gpurun_out/" + f)); r = o["roofline"]; e = o.get("extra", {})
    print(f, "value", round(o["value"], 2), "frac", round(r["frac"], 4), "whole", round(r["whole_token_frac"], 4), "repeats", (e.get("repeats") or {}).get("median"))
    print("  b2b", {k: round(v["ms_per_launch"] * 1e3, 2) for k, v in r["by_kind_back_to_back"].items()})
    print("  prefill", (e.get("prefill") or {}).get("ms"), (e.get("prefill") or {}).get("ms_by_prompt_tokens"), "long", (e.get("long_context") or {}).get("tokens_per_s"),
          "110M", (e.get("stories110M") or {}).get("tokens_per_s"), "42M", (e.get("stories42M") or {}).get("tokens_per_s"), "15M", e.get("stories15M_tokens_per_s"))
    print("  cpu", json.dumps(o.get("cpu_baseline"))[:700])
    print("  solo", json.dumps((e.get("scaling_model") or {}).get("solo_rank"))[:600])
PY
# rocprofv3 kernel table of the SAME bench command (driver's arguments) -- the kernel the roofline object is about -- and of the small shapes
( cd /tmp
rm -rf /tmp/prof_drv
rocprofv3 --kernel-trace --stats -d /tmp/prof_drv -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > /tmp/prof_drv.log 2>&1 || tail -5 /tmp/prof_drv.log
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/prof_drv -name "*.db" | head -1) "round 6 (r06): rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra" > $GRAFT_REPO_ROOT/$O/r06_final_llama2-7b_kernel_stats.md
for wl in stories15M stories42M; do
  rm -rf /tmp/prof_$wl
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$wl -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 64 --warmup 2 --no-cpu-baseline --no-extra > /tmp/prof_$wl.log 2>&1 || tail -5 /tmp/prof_$wl.log
  python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/prof_$wl -name "*.db" | head -1) "round 6 (r06): rocprofv3 --kernel-trace --stats -- python bench.py --workload $wl --steps 64 --warmup 2 --no-cpu-baseline --no-extra" > $GRAFT_REPO_ROOT/$O/r06_final_${wl}_kernel_stats.md
done
for n in 48 64 96 128; do
  rm -rf /tmp/prof_pf$n
  rocprofv3 --kernel-trace --stats -d /tmp/prof_pf$n -o p -- python $GRAFT_REPO_ROOT/scripts/prefill_prof.py llama2-7b $n > /tmp/prof_pf$n.log 2>&1 || tail -5 /tmp/prof_pf$n.log
  python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/prof_pf$n -name "*.db" | head -1) "round 6 (r06): rocprofv3 --kernel-trace --stats -- python scripts/prefill_prof.py llama2-7b $n (3 prefills)" > $GRAFT_REPO_ROOT/$O/r06_final_prefill${n}_llama2-7b.md
done )
head -14 $O/r06_final_llama2-7b_kernel_stats.md
bash scripts/pmc_traffic.sh r06 > $O/r06_final_pmc.log 2>&1; tail -9 $O/r06_final_pmc.log
# the stream form of the bf16-core prefill GEMM: HBM traffic of its four launches by the counters, and its blocks' own clocks
for n in 64 128; do bash scripts/stream_pmc_traffic.sh $n > $O/r06_final_stream_pmc_traffic_$n.md 2>&1; tail -5 $O/r06_final_stream_pmc_traffic_$n.md; done
[ -f llama2.zig_amd/exp/libl2z_x3tl.so ] || bash scripts/x3_timeline.sh > /dev/null 2>&1   # (the measurement build: ~90 s of hipcc)
for n in 64 128; do L2Z_LIB=$PWD/llama2.zig_amd/exp/libl2z_x3tl.so timeout 300 python scripts/x3_timeline.py llama2-7b $n; done > $O/r06_final_stream_timeline.txt 2>&1; grep -E "^[qW]" $O/r06_final_stream_timeline.txt
# perf floors of this tree: two runs (profiles/perf_floor.json holds the worse reading of each)
python scripts/perf_floor.py > $O/r06_final_perf_floor_a.json 2> $O/r06_final_perf_floor_a.err
python scripts/perf_floor.py > $O/r06_final_perf_floor_b.json 2> $O/r06_final_perf_floor_b.err
python -c "
import json
a=json.load(open('$O/r06_final_perf_floor_a.json')); b=json.load(open('$O/r06_final_perf_floor_b.json'))
print('prefill a', {k: round(v, 2) for k, v in a['prefill_ms'].items()}); print('prefill b', {k: round(v, 2) for k, v in b['prefill_ms'].items()})"

timeout 600 python -u scripts/solo_rank.py llama2-7b 128 > $O/r06_solo_rank.md 2>&1; tail -12 $O/r06_solo_rank.md
# two and four ranks on this one GPU, all five legs (a proxy for the control path, the structures' ranking, and the new per-leg diagnostics)
for n in 2 4; do
  L2Z_BENCH_LEG_TIMEOUT_S=300 timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 \
    --master-port 2953$n bench.py --gpus $n --steps 64 --warmup 1 > $O/r06_bench_${n}ranks_1gpu.json 2> $O/r06_bench_${n}ranks_1gpu.err
  echo "$n-rank bench rc=$?"
done
python - <<'PY'
import json
for n in (2, 4):
    try:
        o = json.load(open(f"gpurun_out/r06_bench_{n}ranks_1gpu.json"))
        print(n, "ranks:", round(o.get("value") or 0, 1), o["comm"]["transport"], [(l["transport"], l["ok"], round(l.get("tokens_per_s") or 0, 1), (l.get("cross_device") or {}).get("ll_word_round_trip_us"),
              (l.get("predicted_vs_measured") or {}).get("measured_over_predicted"), l.get("why")) for l in o["comm"]["legs"]])
    except Exception as e:
        print(n, "rank bench line:", e)
PY
( timeout 900 python scripts/fuzz_prefill.py 40 61 wide; timeout 900 python scripts/fuzz_prefill.py 60 62; timeout 600 python scripts/fuzz_shapes.py 80 63; timeout 600 python scripts/fuzz_shapes.py 20 64 wide;
  timeout 600 python scripts/fuzz_shards.py 40 65; timeout 600 python scripts/fuzz_shards.py 30 66 b; timeout 600 python scripts/fuzz_greedy.py 100 67; timeout 600 python scripts/fuzz_hooks.py 150 68 ) > $O/r06_fuzz_full.txt 2>&1
grep -E "^bad:|BAD|ERR" $O/r06_fuzz_full.txt | head -20 > $O/r06_fuzz.txt; cat $O/r06_fuzz.txt
timeout 2400 python -m pytest tests -m gpu -q -rA --durations=8 > $O/r06_final_pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/r06_final_pytest_gpu.log
grep -E "passed|failed|^FAILED|^ERROR" $O/r06_final_pytest_gpu.log | tail -n 8 | tee $O/r06_final_pytest_gpu_tail.txt
grep -E "max \|diff\||max \|logit|identical|margin|vs oracle|vs the stepped|host replay|scheme B|common factor|panel kernel|W2 launch|us per layer|solo rank" $O/r06_final_pytest_gpu.log | head -140 > $O/r06_final_parity_numbers.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
