#!/bin/bash
# round 3, GPU call M: row-sharded prefill on emulated ranks (per-rank time before the exchange), 512 tokens of the
# 7B shape: as the rule picks, and with the two-block form forced on 64 x 64 / 32 x 64 tiles; kernel table at N = 8
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
{
echo "== defaults"; python scripts/sharded_prefill_emu.py llama2-7b 512 2 4 8
echo "== L2Z_PF_KGS=11 (two blocks per 64 x 64 tile everywhere)"; L2Z_PF_KGS=11 python scripts/sharded_prefill_emu.py llama2-7b 512 4 8
echo "== L2Z_PF_KGS=12 (two blocks per 32 x 64 tile everywhere)"; L2Z_PF_KGS=12 python scripts/sharded_prefill_emu.py llama2-7b 512 4 8
echo "== L2Z_PF_KGS=0"; L2Z_PF_KGS=0 python scripts/sharded_prefill_emu.py llama2-7b 512 8
} > $O/r03_sharded_prefill_emu.txt 2>&1
cat $O/r03_sharded_prefill_emu.txt
bash scripts/r2_sharded_prefill_prof.sh 8 r03 > $O/r03_sp8.log 2>&1; head -16 $O/r03_sharded_prefill_w8_kernel_stats.md | cut -c1-150
