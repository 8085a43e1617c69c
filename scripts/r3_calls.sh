#!/bin/bash
# The GPU calls of round 3 other than the final measurement run (scripts/r3_final.sh), as they were made:
#   gpurun -- 'bash scripts/r3_calls.sh <letter>'
# Each section writes under gpurun_out/; what mattered was copied to profiles/ (profiles/README.md says which).
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
case "$1" in
a)
# round 3, GPU call A: the whole gpu-marked suite (new oracle-pinned prefill tests, CLI replay tests, bench legs
# on one GPU), then the N=1 bench line and the 2- and 4-rank legs on the one GPU
timeout 1500 python -m pytest tests -m gpu -q -rA --durations=20 > $O/r03a_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/r03a_pytest_gpu.log
timeout 400 python bench.py > $O/r03a_bench.json 2> $O/r03a_bench.err
for n in 2 4; do
  timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2955$n \
      bench.py --gpus $n --steps 128 --warmup 1 > $O/r03a_mp$n.json 2> $O/r03a_mp$n.err
done
tail -n 5 $O/r03a_pytest_gpu.log
;;
b)
# round 3, GPU call B: head-major KV cache + contiguous split chunks + attention form by position:
# the whole gpu-marked suite, the bench line, interleaved A/B of the short-context form, attention scan
timeout 1500 python -m pytest tests -m gpu -q -rA --durations=8 > $O/r03b_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/r03b_pytest_gpu.log
grep -E "passed|failed" $O/r03b_pytest_gpu.log | tail -n 3
timeout 400 python bench.py > $O/r03b_bench.json 2> $O/r03b_bench.err
{
python scripts/ab.py stories110M 255 5 "" "L2Z_ATTN_SHORT_POS=0" "L2Z_ATTN_SHORT_POS=128"
python scripts/ab.py llama2-7b 255 3 "" "L2Z_ATTN_SHORT_POS=0" "L2Z_ATTN_SHORT_POS=64" "L2Z_ATTN_SHORT_POS=256"
python scripts/attn_time_scan.py llama2-7b 0 63 127 255 256 511 1023 2047
python scripts/attn_time_scan.py stories110M 0 63 127 255 256 1023
} > $O/r03b_ab.txt 2>&1
tail -n 30 $O/r03b_ab.txt
;;
c)
# round 3, GPU call C: rmsnorm inside the short-prompt GEMMs -- parity (prefill tests + fuzz), interleaved A/B
timeout 900 python -m pytest tests -m gpu -q -rA -k "prefill or fuzz or rmsnorm or golden or c_abi" --deselect tests/test_gpu_fullsize.py::test_stories110M_prefill_paths_vs_oracle > $O/r03c_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/r03c_pytest_gpu.log
grep -E "passed|failed" $O/r03c_pytest_gpu.log | tail -n 3
{
for n in 4 16 40; do python scripts/prefill_ab.py llama2-7b $n 5 "" "L2Z_PF_RMS_FUSE=0" "L2Z_PF_ATTN=0" "L2Z_PF_RMS_FUSE=0,L2Z_PF_ATTN=0"; done
python scripts/prefill_ab.py llama2-7b 64 5 "" "L2Z_PF_ATTN=0"
python scripts/prefill_ab.py stories110M 16 5 "" "L2Z_PF_RMS_FUSE=0" "L2Z_PF_ATTN=0"
} > $O/r03c_ab.txt 2>&1
cat $O/r03c_ab.txt
;;
d)
# round 3, GPU call D: split-K family of the tile GEMM -- parity (prefill, sharded, fuzz), interleaved A/B by prompt length
timeout 900 python -m pytest tests -m gpu -q -rA -k "prefill or fuzz or golden or sharded" --deselect tests/test_gpu_fullsize.py::test_stories110M_prefill_paths_vs_oracle > $O/r03d_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/r03d_pytest_gpu.log
grep -E "passed|failed|^FAILED" $O/r03d_pytest_gpu.log | tail -n 12
{
for n in 100 128 200 256; do python scripts/prefill_ab.py llama2-7b $n 4 "L2Z_PF_SPLITK=1" "" "L2Z_PF_SPLITK=2" "L2Z_PF_SPLITK=4"; done
for n in 40 64; do python scripts/prefill_ab.py llama2-7b $n 4 "" "L2Z_PF_SKINNY_MAX=32,L2Z_PF_SPLITK=1" "L2Z_PF_SKINNY_MAX=32,L2Z_PF_SPLITK=2" "L2Z_PF_SKINNY_MAX=32,L2Z_PF_SPLITK=4"; done
python scripts/prefill_ab.py stories110M 128 4 "L2Z_PF_SPLITK=1" "" "L2Z_PF_SPLITK=2" "L2Z_PF_SPLITK=4"
python scripts/prefill_ab.py stories110M 64 4 "" "L2Z_PF_SKINNY_MAX=32,L2Z_PF_SPLITK=1" "L2Z_PF_SKINNY_MAX=32,L2Z_PF_SPLITK=2" "L2Z_PF_SKINNY_MAX=32,L2Z_PF_SPLITK=4"
} > $O/r03d_ab.txt 2>&1
cat $O/r03d_ab.txt
;;
e)
# round 3, GPU call E: the whole gpu suite on the final split-K policy + bench line
timeout 1500 python -m pytest tests -m gpu -q -rA --durations=8 > $O/r03e_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/r03e_pytest_gpu.log
grep -E "passed|failed|^FAILED" $O/r03e_pytest_gpu.log | tail -n 12
timeout 400 python bench.py > $O/r03e_bench.json 2> $O/r03e_bench.err
python scripts/prefill_ab.py llama2-7b 60 4 "" "L2Z_PF_SPLITK=1" > $O/r03e_ab.txt 2>&1
python scripts/prefill_ab.py llama2-7b 100 4 "" "L2Z_PF_SPLITK=1" >> $O/r03e_ab.txt 2>&1
cat $O/r03e_ab.txt
;;
g)
# round 3, GPU call G: is the decode path where round 2 left it?  Round-2 library (built from 86779ae) against the
# current one on the SAME box, alternating processes: per-kind kernel durations and the greedy rate.
{
for i in 1 2; do
  echo "--- round-2 library"; L2Z_LIB=$PWD/scripts/xlib/libllama2_hip_r02.so python scripts/kind_scan.py llama2-7b ""
  echo "--- current library"; python scripts/kind_scan.py llama2-7b ""
done
echo "--- round-2 library"; L2Z_LIB=$PWD/scripts/xlib/libllama2_hip_r02.so python scripts/ab.py llama2-7b 255 3 ""
echo "--- current library"; python scripts/ab.py llama2-7b 255 3 ""
echo "--- round-2 library"; L2Z_LIB=$PWD/scripts/xlib/libllama2_hip_r02.so python scripts/ab.py stories15M 255 3 ""
echo "--- current library"; python scripts/ab.py stories15M 255 3 ""
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -v "^$" | head -30
} > $O/r03_lib_ab.txt 2>&1
cat $O/r03_lib_ab.txt
;;
h)
# round 3, GPU call H: per-launch split-K policy (65-128 tokens: block-starved launches only) -- parity subset + A/B
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
;;
i)
# round 3, GPU call I: regression sweep of every fuzzer after the round's kernel changes (head-major KV cache,
# attention form by position, split-K family, W2 tail); SEED=51 OUT=r03_fuzz_final.txt: again on the final tree
# (split attention: block size by position, at most 8 chunks per head)
S=${SEED:-41}; F=${OUT:-r03_fuzz.txt}
{
echo "== fuzz_shapes 80 (seed $S)"; timeout 600 python scripts/fuzz_shapes.py 80 $S | grep -v "^ok " | tail -n 8
echo "== fuzz_shapes wide 20 (seed $((S+1)))"; timeout 600 python scripts/fuzz_shapes.py 20 $((S+1)) wide | grep -v "^ok " | tail -n 8
echo "== fuzz_shards 80 (seed $S)"; timeout 600 python scripts/fuzz_shards.py 80 $S | grep -v "^ok " | tail -n 8
echo "== fuzz_prefill 160 (seed $S)"; timeout 900 python scripts/fuzz_prefill.py 160 $S | grep -v "^ok " | tail -n 12
echo "== fuzz_greedy 120 (seed $S)"; timeout 600 python scripts/fuzz_greedy.py 120 $S | grep -v "^ok " | tail -n 8
echo "== fuzz_hooks 120 (seed $S)"; timeout 600 python scripts/fuzz_hooks.py 120 $S | grep -v "^ok " | tail -n 8
echo "== fuzz_longctx 6 (seed $S)"; timeout 600 python scripts/fuzz_longctx.py 6 $S | grep -v "^ok " | tail -n 8
echo "== fuzz_p2p 16 (seed $S)"; timeout 900 python scripts/fuzz_p2p.py 16 $S | grep -v "^ok " | tail -n 8
} > $O/$F 2>&1
cat $O/$F
;;
j)
# round 3, GPU call J: bench.py --gpus 8 with all eight ranks on the ONE GPU of the box -- the control path of the
# real thing (24 child processes, three legs, eight gloo ranks), timed end to end
t0=$(date +%s.%N)
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29588 \
    bench.py --gpus 8 --steps 20 --warmup 5 > $O/r03_mp8.json 2> $O/r03_mp8.err
echo "rc=$? wall=$(echo "$(date +%s.%N) - $t0" | bc) s" | tee $O/r03_mp8.txt
python - <<'PY' | tee -a $O/r03_mp8.txt
import json
lines=[l for l in open("$O/r03_mp8.json").read().splitlines() if l.startswith("{")]
d=json.loads(lines[-1])
print("value", d.get("value"), "transport", d.get("comm",{}).get("transport"), "rccl", d.get("comm",{}).get("rccl"))
for l in d["comm"]["legs"]:
    print(" leg", l["transport"], l["ok"], l.get("tokens_per_s"), l.get("why"), "wall", round(l.get("wall_s",0),1), "prefill", (l.get("prefill_sharded") or {}).get("ms"))
PY
tail -n 5 $O/r03_mp8.err
;;
k)
# round 3, GPU call K: the tile GEMM's two k-groups on two blocks (same bits as the unsplit family) -- parity subset,
# then interleaved A/B by prompt length: unsplit | current policy (contiguous split-K) | two-block form by model | forced tiles
timeout 900 python -m pytest tests -m gpu -q -rA -k "prefill or fuzz or sharded" --deselect tests/test_gpu_fullsize.py::test_stories110M_prefill_paths_vs_oracle > $O/r03k_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/r03k_pytest_gpu.log
grep -E "passed|failed|^FAILED" $O/r03k_pytest_gpu.log | tail -n 8
{
U="L2Z_PF_SPLITK=1"
for n in 100 128 200 256 300 512; do
  python scripts/prefill_ab.py llama2-7b $n 3 "$U,L2Z_PF_KGS=0" "L2Z_PF_KGS=0" "$U" "$U,L2Z_PF_KGS=10" "$U,L2Z_PF_KGS=11" "$U,L2Z_PF_KGS=14"
done
S="L2Z_PF_SKINNY_MAX=32,L2Z_PF_SPLITK=1"
python scripts/prefill_ab.py llama2-7b 64 3 "$S,L2Z_PF_KGS=0" "L2Z_PF_KGS=0" "$S" "$S,L2Z_PF_KGS=11" "$S,L2Z_PF_KGS=12"
python scripts/prefill_ab.py stories110M 128 3 "$U,L2Z_PF_KGS=0" "$U" "$U,L2Z_PF_KGS=11" "$U,L2Z_PF_KGS=12"
python scripts/prefill_ab.py stories110M 512 3 "$U,L2Z_PF_KGS=0" "$U" "$U,L2Z_PF_KGS=11"
} > $O/r03k_ab.txt 2>&1
cat $O/r03k_ab.txt
;;
l)
# round 3, GPU call L: two-block form at long chunks (512 / 1024 / 2000 tokens), interleaved
{
python scripts/prefill_ab.py llama2-7b 512 3 "L2Z_PF_KGS=0" "" "L2Z_PF_KGS=10"
python scripts/prefill_ab.py llama2-7b 1024 3 "L2Z_PF_KGS=0" "" "L2Z_PF_KGS=10" "L2Z_PF_KGS=14"
python scripts/prefill_ab.py llama2-7b 2000 3 "L2Z_PF_KGS=0" ""
python scripts/prefill_ab.py llama2-7b 700 3 "L2Z_PF_KGS=0" ""
} > $O/r03l_ab.txt 2>&1
cat $O/r03l_ab.txt
;;
m)
# round 3, GPU call M: row-sharded prefill on emulated ranks (per-rank time before the exchange), 512 tokens of the
# 7B shape: as the rule picks, and with the two-block form forced on 64 x 64 / 32 x 64 tiles; kernel table at N = 8
{
echo "== defaults"; python scripts/sharded_prefill_emu.py llama2-7b 512 2 4 8
echo "== L2Z_PF_KGS=11 (two blocks per 64 x 64 tile everywhere)"; L2Z_PF_KGS=11 python scripts/sharded_prefill_emu.py llama2-7b 512 4 8
echo "== L2Z_PF_KGS=12 (two blocks per 32 x 64 tile everywhere)"; L2Z_PF_KGS=12 python scripts/sharded_prefill_emu.py llama2-7b 512 4 8
echo "== L2Z_PF_KGS=0"; L2Z_PF_KGS=0 python scripts/sharded_prefill_emu.py llama2-7b 512 8
} > $O/r03_sharded_prefill_emu.txt 2>&1
cat $O/r03_sharded_prefill_emu.txt
bash scripts/r2_sharded_prefill_prof.sh 8 r03 > $O/r03_sp8.log 2>&1; head -16 $O/r03_sharded_prefill_w8_kernel_stats.md | cut -c1-150
;;
n)
# round 3, GPU call N: the fused qkv + attention launch of small MHA models with 8 lanes x 9 float4 per row pair (all
# of a head's 72 pairs requested in one pass; measured slower and reverted, like the late-V form before it: git log)
# -- parity (fused tests, 15M greedy ids), interleaved A/B
timeout 900 python -m pytest tests -m gpu -q -rA -k "fused or stories15M or greedy_token or golden or fuzz" > $O/r03n_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/r03n_pytest_gpu.log
grep -E "passed|failed|^FAILED" $O/r03n_pytest_gpu.log | tail -n 8
{
python scripts/ab.py stories15M 255 7 "" "L2Z_FUSE_SMALL_LPR8=0" "L2Z_FUSE_SMALL=0"
python scripts/kind_scan.py stories15M "" "L2Z_FUSE_SMALL_LPR8=0"
} > $O/r03_fused_lpr8_ab.txt 2>&1
cat $O/r03_fused_lpr8_ab.txt
;;
o)
# round 3, GPU call O: the narrow-row mat-vec's grid -- resident blocks per CU assumed (L2Z_MV_OCC) instead of the
# occupancy query's answer (rocprofv3 shows 500 blocks for the 15M classifier's 4000 units: two units per wave).
# Needs the L2Z_MV_OCC knob of commit-time only (git log: measured, more blocks are slower, knob removed again).
{
python scripts/kind_scan.py stories15M "" "L2Z_MV_OCC=4" "L2Z_MV_OCC=8"
python scripts/kind_scan.py stories110M "" "L2Z_MV_OCC=4" "L2Z_MV_OCC=8"
python scripts/ab.py stories15M 255 5 "" "L2Z_MV_OCC=4" "L2Z_MV_OCC=8"
python scripts/ab.py stories110M 255 5 "" "L2Z_MV_OCC=4" "L2Z_MV_OCC=8"
} > $O/r03_mv_occ_ab.txt 2>&1
cat $O/r03_mv_occ_ab.txt
;;
p)
# round 3, GPU call P: split attention with 256 threads per block below pos 1024 (profiles/r03_attn_split_scan.txt) --
# parity (attention kernels, split tests, 110M across the switch-overs, 7B sharded positions), interleaved A/B
timeout 900 python -m pytest tests -m gpu -q -rA -k "attention or split or across or 7b_sharded or fuzz_longctx or random_shapes" > $O/r03p_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/r03p_pytest_gpu.log
grep -E "passed|failed|^FAILED" $O/r03p_pytest_gpu.log | tail -n 8
{
python scripts/ab.py llama2-7b 64 3 300 "" "L2Z_ATTN_SPLIT_WIDE_POS=0"
python scripts/ab.py llama2-7b 64 3 700 "" "L2Z_ATTN_SPLIT_WIDE_POS=0"
python scripts/ab.py llama2-7b 32 3 1960 "" "L2Z_ATTN_SPLIT_WIDE_POS=0"
python scripts/ab.py stories110M 255 3 300 "" "L2Z_ATTN_SPLIT_WIDE_POS=0"
python scripts/attn_time_scan.py llama2-7b 255 256 511 1023 1024 2047
timeout 300 python scripts/fuzz_longctx.py 6 43 | tail -n 3
} > $O/r03_attn_split_nt_ab.txt 2>&1
cat $O/r03_attn_split_nt_ab.txt
;;
q)
# round 3, GPU call Q (and the ones after it): the K-sliced short-prompt GEMM with X in registers (commit 93b3cc1, removed
# again in the next one) against the short-prompt kernels, and scripts/dma_pattern_probe -- profiles/r03_dma_pattern_probe.txt
for i in $(seq 0 22); do timeout 60 ./scripts/dma_pattern_probe $i 2>&1 | tail -n 1; done > $O/r03_dma_pattern_probe.txt
cat $O/r03_dma_pattern_probe.txt
;;
*) echo "usage: r3_calls.sh a|b|c|d|e|g|h|i|j|k|l|m|n|o|p|q"; exit 2;;
esac
