#!/bin/bash
# The GPU calls of round 4, as they were made:   gpurun -- 'bash scripts/r4_calls.sh <letter>'
# Each section writes under gpurun_out/; what mattered was copied to profiles/ (profiles/README.md says which).
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
case "$1" in
a)
# round 4, GPU call A: the overlapped decode chain (two streams, LL hand-overs, duo mat-vecs) -- does it run, are
# the bits those of the single chain, and what does it buy: interleaved A/B at the 7B shape by mode, edge set and
# hint form; then the decode parity tests
export L2Z_P2P_TIMEOUT_S=3
{
timeout 300 python scripts/ab.py llama2-7b 128 3 "" "L2Z_OVERLAP=0" "L2Z_DUO=0" "L2Z_NO_GRAPH=1" "L2Z_NO_GRAPH=1,L2Z_OVERLAP=0"
echo "rc=$?"
timeout 300 python scripts/ab.py llama2-7b 128 3 "" "L2Z_OVERLAP_EDGES=14" "L2Z_OVERLAP_EDGES=13" "L2Z_OVERLAP_EDGES=11" "L2Z_OVERLAP_EDGES=7" "L2Z_OVERLAP_EDGES=8" "L2Z_OVERLAP_HINT=0" "L2Z_OVERLAP_HINT_SLEEP=1" "L2Z_OVERLAP_HINT_SLEEP=6"
echo "rc=$?"
timeout 300 python scripts/ab.py llama2-7b 64 3 300 "" "L2Z_OVERLAP=0" "L2Z_DUO=0"
echo "rc=$?"
timeout 300 python scripts/ab.py llama2-7b 48 3 1900 "" "L2Z_OVERLAP=0" "L2Z_DUO=0"
echo "rc=$?"
} > $O/r04a_ab.txt 2>&1
cat $O/r04a_ab.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "transformer or greedy or golden or c_abi or attention or 7b" > $O/r04a_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/r04a_pytest_gpu.log
tail -n 15 $O/r04a_pytest_gpu.log
;;
b)
# round 4, GPU call B: WHERE the overlapped chain loses -- rocprofv3 kernel timelines (start / end per launch and
# queue) of four modes, per-kind durations duo vs row kernel, and the polling period
export L2Z_P2P_TIMEOUT_S=3
cd /tmp
i=0
for mode in "" "L2Z_OVERLAP_EDGES=8" "L2Z_OVERLAP=0" "L2Z_DUO=0"; do
  i=$((i+1))
  rm -rf /tmp/tl_$i
  env $mode rocprofv3 --kernel-trace -d /tmp/tl_$i -o t -- python $GRAFT_REPO_ROOT/scripts/decode_steps.py llama2-7b 12 > /tmp/tl_$i.log 2>&1 || tail -5 /tmp/tl_$i.log
  tail -n 1 /tmp/tl_$i.log
  python $GRAFT_REPO_ROOT/scripts/timeline_report.py $(find /tmp/tl_$i -name "*.db" | head -1) "round 4 (r04b): [${mode:-defaults}] rocprofv3 --kernel-trace -- python scripts/decode_steps.py llama2-7b 12" > $GRAFT_REPO_ROOT/$O/r04b_timeline_$i.md 2>&1
done
cd $GRAFT_REPO_ROOT
cat $O/r04b_timeline_*.md
{
timeout 200 python scripts/kind_ab.py llama2-7b 8 "" "L2Z_DUO=0"
timeout 300 python scripts/ab.py llama2-7b 128 3 "L2Z_OVERLAP_HINT_SLEEP=6" "L2Z_OVERLAP_HINT_SLEEP=32" "L2Z_OVERLAP_EDGES=8,L2Z_OVERLAP_HINT_SLEEP=6" "L2Z_OVERLAP_EDGES=8,L2Z_OVERLAP_HINT_SLEEP=32" "L2Z_OVERLAP_EDGES=8,L2Z_OVERLAP_HINT_SLEEP=200"
} > $O/r04b_ab.txt 2>&1
cat $O/r04b_ab.txt
;;
c)
# round 4, GPU call C: (1) what a merely RESIDENT kernel in another HW queue costs a chain of streaming launches
# (scripts/coresident_probe.hip); (2) the duo mat-vec without the conditional consume: per-kind durations and A/B
hipcc --offload-arch=gfx950 -O3 -o /tmp/coresident_probe scripts/coresident_probe.hip && timeout 120 /tmp/coresident_probe > $O/r04c_coresident_probe.txt 2>&1
cat $O/r04c_coresident_probe.txt
export L2Z_P2P_TIMEOUT_S=3
{
timeout 200 python scripts/kind_ab.py llama2-7b 8 "" "L2Z_DUO=0"
timeout 300 python scripts/ab.py llama2-7b 128 3 "" "L2Z_OVERLAP=0" "L2Z_DUO=0"
} > $O/r04c_ab.txt 2>&1
cat $O/r04c_ab.txt
;;
d)
# round 4, GPU call D: duo staging with x requested ahead of the weights; hint one sweep early; paced run-ahead
export L2Z_P2P_TIMEOUT_S=3
{
timeout 200 python scripts/kind_ab.py llama2-7b 8 "" "L2Z_DUO=0"
timeout 400 python scripts/ab.py llama2-7b 128 3 "" "L2Z_OVERLAP_DEFER=0" "L2Z_OVERLAP_HINT_BACK=0" "L2Z_OVERLAP_DEFER=0,L2Z_OVERLAP_HINT_BACK=0" "L2Z_OVERLAP_HINT_BACK=2" "L2Z_OVERLAP_EDGES=1" "L2Z_OVERLAP_EDGES=9" "L2Z_OVERLAP=0" "L2Z_DUO=0"
} > $O/r04d_ab.txt 2>&1
cat $O/r04d_ab.txt
;;
e)
# round 4, GPU call E: hand-over once per block (outputs wait in LDS) instead of LL stores after every unit
export L2Z_P2P_TIMEOUT_S=3
{
timeout 400 python scripts/ab.py llama2-7b 128 3 "" "L2Z_OVERLAP_EDGES=1" "L2Z_OVERLAP_EDGES=8" "L2Z_OVERLAP_EDGES=14" "L2Z_OVERLAP_HINT_SLEEP=1" "L2Z_OVERLAP=0" "L2Z_DUO=0"
} > $O/r04e_ab.txt 2>&1
cat $O/r04e_ab.txt
;;
f)
# round 4, GPU call F: the hand-over as the GPU's own clock sees it (measurement build with stamps in the duo kernel)
export L2Z_P2P_TIMEOUT_S=3 L2Z_LIB=$PWD/llama2.zig_amd/libllama2_hip_tl.so L2Z_NO_GRAPH=1
{
for mode in "L2Z_OVERLAP=1" "L2Z_OVERLAP_EDGES=8" "L2Z_OVERLAP_EDGES=1" "L2Z_OVERLAP=0"; do
  env $mode timeout 200 python scripts/decode_timeline.py llama2-7b 8
done
} > $O/r04f_timeline.txt 2>&1
cat $O/r04f_timeline.txt
;;
g)
# round 4, GPU call G: hint = 16 producer blocks' words, LL residual requested once per block at entry
export L2Z_P2P_TIMEOUT_S=3
{
timeout 400 python scripts/ab.py llama2-7b 128 3 "" "L2Z_OVERLAP_EDGES=1" "L2Z_OVERLAP_EDGES=8" "L2Z_OVERLAP_EDGES=11" "L2Z_OVERLAP=0" "L2Z_DUO=0"
for mode in "L2Z_OVERLAP=1" "L2Z_OVERLAP=0"; do
  env $mode L2Z_LIB=$PWD/llama2.zig_amd/libllama2_hip_tl.so L2Z_NO_GRAPH=1 timeout 200 python scripts/decode_timeline.py llama2-7b 8
done
} > $O/r04g_ab.txt 2>&1
cat $O/r04g_ab.txt
;;
h)
# round 4, GPU call H: the timeline again with stamps that do not perturb (registers, stored at block end)
export L2Z_P2P_TIMEOUT_S=3
{
for mode in "L2Z_OVERLAP=1" "L2Z_OVERLAP_EDGES=1" "L2Z_OVERLAP=0"; do
  env $mode L2Z_LIB=$PWD/llama2.zig_amd/libllama2_hip_tl.so L2Z_NO_GRAPH=1 timeout 200 python scripts/decode_timeline.py llama2-7b 6
done
} > $O/r04h_timeline.txt 2>&1
cat $O/r04h_timeline.txt
;;
i)
# round 4, GPU call I: the per-rank cost model of a sharded decode token (emulated ranks), process-to-process variance
# of the headline, the whole gpu-marked suite (KV tolerance 2e-5, ADVICE fixes), the bench line with repeats + scaling_model
timeout 300 python scripts/sharded_decode_emu.py 8 1024 > $O/r04_sharded_decode_emu.md 2>&1
cat $O/r04_sharded_decode_emu.md
{ for i in 1 2 3 4 5 6; do timeout 120 python scripts/decode_steps.py llama2-7b 255; done; } > $O/r04i_process_variance.txt 2>&1
cat $O/r04i_process_variance.txt
timeout 1500 python -m pytest tests -m gpu -q -rA --durations=8 > $O/r04i_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/r04i_pytest_gpu.log
grep -E "passed|failed|^FAILED|rc=" $O/r04i_pytest_gpu.log | tail -n 8
timeout 600 python bench.py > $O/r04i_bench.json 2> $O/r04i_bench.err
tail -c 1500 $O/r04i_bench.json; tail -n 3 $O/r04i_bench.err
;;
j)
# round 4, GPU call J: the persistent decode launches (engine.hip): bits and speed against the launch chain
export L2Z_P2P_TIMEOUT_S=3
{
timeout 300 python scripts/ab.py llama2-7b 128 3 "" "L2Z_ENGINE=1"
echo "rc=$?"
timeout 300 python scripts/ab.py llama2-7b 64 3 300 "" "L2Z_ENGINE=1"
echo "rc=$?"
} > $O/r04j_ab.txt 2>&1
cat $O/r04j_ab.txt
;;
k)
# round 4, GPU call K: where a persistent launch spends its time (measurement build)
export L2Z_P2P_TIMEOUT_S=3
L2Z_LIB=$PWD/llama2.zig_amd/libllama2_hip_tl.so L2Z_NO_GRAPH=1 L2Z_ENGINE=1 timeout 200 python scripts/engine_timeline.py 4 > $O/r04k_engine_timeline.md 2>&1
cat $O/r04k_engine_timeline.md
;;
l)
# round 4, GPU call L: four gatherer waves -- bits, speed, timeline
export L2Z_P2P_TIMEOUT_S=3
{
timeout 300 python scripts/ab.py llama2-7b 128 3 "" "L2Z_ENGINE=1"
echo "rc=$?"
L2Z_LIB=$PWD/llama2.zig_amd/libllama2_hip_tl.so L2Z_NO_GRAPH=1 L2Z_ENGINE=1 timeout 200 python scripts/engine_timeline.py 4
} > $O/r04l_engine.txt 2>&1
cat $O/r04l_engine.txt
;;
m)
# debug: one engine pass with a short timeout
L2Z_P2P_TIMEOUT_S=1 L2Z_ENGINE=1 L2Z_NO_GRAPH=1 timeout 100 python scripts/decode_steps.py llama2-7b 2 > $O/r04m_dbg.txt 2>&1
tail -5 $O/r04m_dbg.txt
;;
n)
# round 4, GPU call N: the opt-in forms' test; the engine after the start-order fix
export L2Z_P2P_TIMEOUT_S=3
timeout 600 python -m pytest tests/test_gpu_chain_forms.py -m gpu -q -x 2>&1 | tail -5
timeout 200 python scripts/ab.py llama2-7b 128 3 "" "L2Z_ENGINE=1" 2>&1 | tail -3
;;
p)
# round 4, GPU call P: the persistent launches on a shard group (2 and 4 processes on the one GPU), bit for bit
timeout 900 python -m pytest tests/test_gpu_p2p.py -m gpu -q -x -k "engine or (row-kernel and x2-consume)" 2>&1 | tail -15
;;
q)
timeout 600 python -m pytest tests/test_gpu_chain_forms.py -m gpu -q -k "engine" 2>&1 | tail -12
;;
r)
timeout 200 python scripts/dbg_engine_shard.py 2 2>&1 | tail -6
DBG_ENGINE=0 timeout 200 python scripts/dbg_engine_shard.py 2 2>&1 | tail -3
;;
s)
# round 4, GPU call S: the forms' tests with the runstate-form assertion, the peer-write tests with the fourth bench leg,
# then bench.py --gpus 4 on the 7B shape with all four ranks on this one GPU (a proxy: the kernels share the chip)
export L2Z_P2P_TIMEOUT_S=20
timeout 600 python -m pytest tests/test_gpu_chain_forms.py -m gpu -q -x 2>&1 | tail -5
timeout 1200 python -m pytest tests/test_gpu_p2p.py -m gpu -q -x 2>&1 | tail -8
L2Z_BENCH_LEG_TIMEOUT_S=240 timeout 1100 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 \
  --master-port 29533 bench.py --gpus 4 --steps 64 --warmup 1 > $O/r04_bench_4ranks_1gpu.json 2> $O/r04_bench_4ranks_1gpu.err
echo "bench rc=$?"; tail -c 1500 $O/r04_bench_4ranks_1gpu.err
python - <<'PY'
import json
for ln in open("gpurun_out/r04_bench_4ranks_1gpu.json"):
    if ln.startswith("{"):
        o = json.loads(ln)
        print(o.get("value"), o.get("comm", {}).get("transport"))
        for l in o["comm"]["legs"]:
            print(l["transport"], l["ok"], l.get("tokens_per_s"), l.get("why"), l.get("runstate_form"), l.get("wall_s"))
PY
;;
t)
# round 4, GPU call T: scheme B (column-sharded Wo / W2, all-reduces) -- emulated ranks, then real processes, then the legs
export L2Z_P2P_TIMEOUT_S=20
timeout 900 python -m pytest tests/test_gpu_scheme_b.py -m gpu -q -x -s 2>&1 | tail -25
timeout 900 python -m pytest tests/test_gpu_p2p.py -m gpu -q -x -s -k "scheme_b or bench_legs" 2>&1 | tail -25
;;
u)
# round 4, GPU call U: scheme B after the parallel-poll reduce kernel, the 1-rank RCCL all-reduce path, the CLI under scheme B;
# then bench.py --gpus 4 (7B shape, four ranks on this one GPU: a proxy) with all six legs
export L2Z_P2P_TIMEOUT_S=20
timeout 900 python -m pytest tests/test_gpu_scheme_b.py -m gpu -q -x 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_p2p.py tests/test_host_cli.py -m gpu -q -x -k "scheme_b or bench_legs" 2>&1 | tail -8
L2Z_BENCH_LEG_TIMEOUT_S=240 timeout 1100 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 \
  --master-port 29533 bench.py --gpus 4 --steps 64 --warmup 1 > $O/r04_bench_4ranks_1gpu.json 2> $O/r04_bench_4ranks_1gpu.err
echo "bench rc=$?"; tail -c 600 $O/r04_bench_4ranks_1gpu.err
python - <<'PY'
import json
for ln in open("gpurun_out/r04_bench_4ranks_1gpu.json"):
    if ln.startswith("{"):
        o = json.loads(ln)
        print(o.get("value"), o.get("comm", {}).get("transport"), o.get("comm", {}).get("scheme_b"))
        for l in o["comm"]["legs"]:
            print(l["transport"], l.get("scheme"), l["ok"], l.get("tokens_per_s"), l.get("why"), l.get("runstate_form"), l.get("wall_s"), l.get("us_per_gather"))
PY
;;
v)
# round 4, GPU call V: the split attention kernel at pos 2047 (and 1023, 4095-equivalents) by its blocks' own clocks
for pos in 2047 1023 511; do
L2Z_LIB=$PWD/llama2.zig_amd/libllama2_hip_tl.so L2Z_NO_GRAPH=1 timeout 300 python scripts/attn_timeline.py llama2-7b $pos 4
done > $O/r04_attn_timeline.md 2>&1
cat $O/r04_attn_timeline.md
;;
w)
# round 4, GPU call W: synchronisation and streaming inside ONE XCD (scripts/xcd_local_probe.hip)
hipcc --offload-arch=gfx950 -O3 -o /tmp/xcd_local_probe scripts/xcd_local_probe.hip && timeout 120 /tmp/xcd_local_probe > $O/r04_xcd_local_probe.txt 2>&1
cat $O/r04_xcd_local_probe.txt
;;
x)
# round 4, GPU call X: the split attention kernel after the change of row ownership (pieces dealt to the chunks, rows below the
# form's first position requested before pos has arrived) and the one-round-trip combine: parity tests, then the timeline
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x -k "attention or attn or split or sharded_hip or long" 2>&1 | tail -5
for pos in 2047 1023 511 300; do
L2Z_LIB=$PWD/llama2.zig_amd/libllama2_hip_tl.so L2Z_NO_GRAPH=1 timeout 300 python scripts/attn_timeline.py llama2-7b $pos 4
done > $O/r04_attn_timeline_2.md 2>&1
cat $O/r04_attn_timeline_2.md
;;
y)
# round 4, GPU call Y: regression sweep of every fuzzer on the round's tree (scheme B included), fresh seeds
S=${SEED:-61}; F=${OUT:-r04_fuzz.txt}
{
echo "== fuzz_shapes 80 (seed $S)"; timeout 600 python scripts/fuzz_shapes.py 80 $S | grep -v "^ok " | tail -n 8
echo "== fuzz_shapes wide 20 (seed $((S+1)))"; timeout 600 python scripts/fuzz_shapes.py 20 $((S+1)) wide | grep -v "^ok " | tail -n 8
echo "== fuzz_shards 80 (seed $S)"; timeout 600 python scripts/fuzz_shards.py 80 $S | grep -v "^ok " | tail -n 8
echo "== fuzz_shards scheme B 80 (seed $((S+2)))"; timeout 600 python scripts/fuzz_shards.py 80 $((S+2)) b | grep -v "^ok " | tail -n 8
echo "== fuzz_prefill 160 (seed $S)"; timeout 900 python scripts/fuzz_prefill.py 160 $S | grep -v "^ok " | tail -n 12
echo "== fuzz_greedy 120 (seed $S)"; timeout 600 python scripts/fuzz_greedy.py 120 $S | grep -v "^ok " | tail -n 8
echo "== fuzz_hooks 120 (seed $S)"; timeout 600 python scripts/fuzz_hooks.py 120 $S | grep -v "^ok " | tail -n 8
echo "== fuzz_longctx 6 (seed $S)"; timeout 600 python scripts/fuzz_longctx.py 6 $S | grep -v "^ok " | tail -n 8
echo "== fuzz_p2p 16 (seed $S)"; timeout 900 python scripts/fuzz_p2p.py 16 $S | grep -v "^ok " | tail -n 8
} > $O/$F 2>&1
cat $O/$F
;;
z)
# round 4, GPU call Z: rocprofv3 kernel tables of ONE rank of 8 alone on the GPU (solo connect), whole pass, per structure
export L2Z_P2P_TIMEOUT_S=5
( cd /tmp
for k in ${FORMS:-0 3}; do
  rm -rf /tmp/prof_solo$k
  rocprofv3 --kernel-trace --stats -d /tmp/prof_solo$k -o p -- python $GRAFT_REPO_ROOT/scripts/solo_rank.py llama2-7b 12 8 $k > /tmp/prof_solo$k.log 2>&1 || tail -5 /tmp/prof_solo$k.log
  grep "tok/s" /tmp/prof_solo$k.log
  python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/prof_solo$k -name "*.db" | head -1) "round 4: rocprofv3 --kernel-trace --stats -- python scripts/solo_rank.py llama2-7b 12 8 $k (one rank of 8 alone, hand-overs free)" > $GRAFT_REPO_ROOT/$O/r04_solo8_form${k}_kernel_stats.md
  head -16 $GRAFT_REPO_ROOT/$O/r04_solo8_form${k}_kernel_stats.md | cut -c1-150
done )
;;
esac
