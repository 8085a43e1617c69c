#!/bin/bash
# round 5, sixth GPU call: fuzz sweeps over the new paths (wide prefill shapes: panel kernel, scheme B; real processes)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python scripts/fuzz_prefill.py 60 31 wide > $O/r05_fuzz_prefill_wide.txt 2>&1; tail -3 $O/r05_fuzz_prefill_wide.txt; grep -c "^ok" $O/r05_fuzz_prefill_wide.txt; grep -v "^ok" $O/r05_fuzz_prefill_wide.txt | head -20
timeout 600 python scripts/fuzz_prefill.py 80 32 > $O/r05_fuzz_prefill.txt 2>&1; tail -2 $O/r05_fuzz_prefill.txt; grep -v "^ok" $O/r05_fuzz_prefill.txt | head
timeout 600 python scripts/fuzz_p2p.py 16 33 > $O/r05_fuzz_p2p.txt 2>&1; tail -2 $O/r05_fuzz_p2p.txt; grep -v "^ok" $O/r05_fuzz_p2p.txt | head
timeout 400 python scripts/fuzz_shards.py 40 34 > $O/r05_fuzz_shards.txt 2>&1; tail -2 $O/r05_fuzz_shards.txt
timeout 400 python scripts/fuzz_greedy.py 120 35 > $O/r05_fuzz_greedy.txt 2>&1; tail -2 $O/r05_fuzz_greedy.txt
timeout 300 python -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -3
