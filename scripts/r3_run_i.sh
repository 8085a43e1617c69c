#!/bin/bash
# round 3, GPU call I: regression sweep of every fuzzer after the round's kernel changes (head-major KV cache,
# attention form by position, split-K family, W2 tail)
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
{
echo "== fuzz_shapes 80 (seed 41)"; timeout 600 python scripts/fuzz_shapes.py 80 41 | grep -v "^ok " | tail -n 8
echo "== fuzz_shapes wide 20 (seed 42)"; timeout 600 python scripts/fuzz_shapes.py 20 42 wide | grep -v "^ok " | tail -n 8
echo "== fuzz_shards 80 (seed 41)"; timeout 600 python scripts/fuzz_shards.py 80 41 | grep -v "^ok " | tail -n 8
echo "== fuzz_prefill 160 (seed 41)"; timeout 900 python scripts/fuzz_prefill.py 160 41 | grep -v "^ok " | tail -n 12
echo "== fuzz_greedy 120 (seed 41)"; timeout 600 python scripts/fuzz_greedy.py 120 41 | grep -v "^ok " | tail -n 8
echo "== fuzz_hooks 120 (seed 41)"; timeout 600 python scripts/fuzz_hooks.py 120 41 | grep -v "^ok " | tail -n 8
echo "== fuzz_longctx 6 (seed 41)"; timeout 600 python scripts/fuzz_longctx.py 6 41 | grep -v "^ok " | tail -n 8
echo "== fuzz_p2p 16 (seed 41)"; timeout 900 python scripts/fuzz_p2p.py 16 41 | grep -v "^ok " | tail -n 8
} > $O/r03_fuzz.txt 2>&1
cat $O/r03_fuzz.txt
