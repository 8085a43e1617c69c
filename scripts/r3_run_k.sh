#!/bin/bash
# round 3, GPU call K: the tile GEMM's two k-groups on two blocks (same bits as the unsplit family) -- parity subset,
# then interleaved A/B by prompt length: unsplit | current policy (contiguous split-K) | two-block form by model | forced tiles
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
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
